/*
 * hammock_jni.c -- JNI glue between cz.krejciadam.hammock.HipNative and the C ABI of
 * include/hammock_hip.h.  SOURCE ONLY: this image has no JDK (no jni.h), so this file is
 * not part of the in-tree build; tests/test_cli_io.py compiles it against a declaration-only
 * stand-in (tests/jni_stub/jni.h) so that it cannot rot.  Build where a JDK exists:
 *
 *   gcc -shared -fPIC -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -Iinclude \
 *       hammock_amd/java/jni/hammock_jni.c -Lhammock_amd/lib -lhammock_hip \
 *       -Wl,-rpath,'$ORIGIN' -o hammock_amd/lib/libhammock_jni.so
 */
#include <jni.h>
#include <stdint.h>

#include <stddef.h>

#include "hammock_hip.h"

static void throw_for(JNIEnv *env, hmk_ctx *ctx, int st) {
    const char *msg = hmk_last_error(ctx);
    const char *cls;
    switch (st) {
        case HMK_ERR_SHIFT_TOO_BIG: cls = "cz/krejciadam/hammock/DataException"; break;          /* ShiftedScorer.java:59-62 */
        case HMK_ERR_REFERENCE_WOULD_CRASH: cls = "java/lang/NullPointerException"; break;        /* Hammock.java:153-157 */
        case HMK_ERR_BAD_ARG: case HMK_ERR_NO_SEQUENCES: cls = "java/lang/IllegalArgumentException"; break;
        case HMK_ERR_OOM: cls = "java/lang/OutOfMemoryError"; break;
        default: cls = "java/lang/IllegalStateException"; break;                                 /* HMK_ERR_DEVICE ... */
    }
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), msg);
}

JNIEXPORT jlong JNICALL Java_cz_krejciadam_hammock_HipNative_create(JNIEnv *env, jclass c, jintArray matrix, jint device) {
    (void)c;
    hmk_ctx *ctx = NULL;
    jint *m = (*env)->GetIntArrayElements(env, matrix, NULL);
    int st = hmk_create((const int32_t *)m, device, &ctx);
    (*env)->ReleaseIntArrayElements(env, matrix, m, JNI_ABORT);
    if (st) throw_for(env, NULL, st);
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT jlong JNICALL Java_cz_krejciadam_hammock_HipNative_createMulti(JNIEnv *env, jclass c, jintArray matrix, jintArray devices) {
    (void)c;
    hmk_ctx *ctx = NULL;
    jsize n = (*env)->GetArrayLength(env, devices);
    jint *m = (*env)->GetIntArrayElements(env, matrix, NULL);
    jint *d = (*env)->GetIntArrayElements(env, devices, NULL);
    int st = hmk_create_multi((const int32_t *)m, (const int *)d, (int)n, &ctx);
    (*env)->ReleaseIntArrayElements(env, matrix, m, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, devices, d, JNI_ABORT);
    if (st) throw_for(env, NULL, st);
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_cz_krejciadam_hammock_HipNative_destroy(JNIEnv *env, jclass c, jlong ctx) {
    (void)env; (void)c;
    hmk_destroy((hmk_ctx *)(intptr_t)ctx);
}

JNIEXPORT void JNICALL Java_cz_krejciadam_hammock_HipNative_reserve(JNIEnv *env, jclass c, jlong h, jint n) {
    (void)c;
    int st = hmk_reserve((hmk_ctx *)(intptr_t)h, (uint32_t)n);
    if (st) throw_for(env, (hmk_ctx *)(intptr_t)h, st);
}

JNIEXPORT void JNICALL Java_cz_krejciadam_hammock_HipNative_setJavaHashset(JNIEnv *env, jclass c, jlong h, jint version) {
    (void)c;
    int st = hmk_set_java_hashset((hmk_ctx *)(intptr_t)h, (int)version);
    if (st) throw_for(env, (hmk_ctx *)(intptr_t)h, st);
}

JNIEXPORT void JNICALL Java_cz_krejciadam_hammock_HipNative_setSequences(JNIEnv *env, jclass c, jlong h, jbyteArray residues,
                                                                         jintArray offsets, jintArray sizes) {
    (void)c;
    hmk_ctx *ctx = (hmk_ctx *)(intptr_t)h;
    jsize n = (*env)->GetArrayLength(env, sizes);
    jbyte *r = (*env)->GetByteArrayElements(env, residues, NULL);
    jint *o = (*env)->GetIntArrayElements(env, offsets, NULL);
    jint *s = (*env)->GetIntArrayElements(env, sizes, NULL);
    int st = hmk_set_sequences(ctx, (const uint8_t *)r, (const uint32_t *)o, (const int32_t *)s, (uint32_t)n);
    (*env)->ReleaseByteArrayElements(env, residues, r, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, offsets, o, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, sizes, s, JNI_ABORT);
    if (st) throw_for(env, ctx, st);
}

JNIEXPORT jintArray JNICALL Java_cz_krejciadam_hammock_HipNative_scoreWithShift(JNIEnv *env, jclass c, jlong h, jint i, jint j,
                                                                                jint maxShift, jint shiftPenalty) {
    (void)c;
    hmk_ctx *ctx = (hmk_ctx *)(intptr_t)h;
    uint32_t ui = (uint32_t)i, uj = (uint32_t)j;
    int32_t out[2] = {0, 0};
    int st = hmk_score_with_shift(ctx, &ui, &uj, 1, maxShift, shiftPenalty, &out[0], &out[1]);
    if (st) { throw_for(env, ctx, st); return NULL; }
    jintArray res = (*env)->NewIntArray(env, 2);
    (*env)->SetIntArrayRegion(env, res, 0, 2, (const jint *)out);
    return res;
}

JNIEXPORT jint JNICALL Java_cz_krejciadam_hammock_HipNative_scoreLocal(JNIEnv *env, jclass c, jlong h, jint i, jint j,
                                                                       jint gapOpen, jint gapExtend) {
    (void)c;
    hmk_ctx *ctx = (hmk_ctx *)(intptr_t)h;
    uint32_t ui = (uint32_t)i, uj = (uint32_t)j;
    int32_t out = 0;
    int st = hmk_score_pairs_local(ctx, &ui, &uj, 1, gapOpen, gapExtend, &out);
    if (st) throw_for(env, ctx, st);
    return out;
}

JNIEXPORT jint JNICALL Java_cz_krejciadam_hammock_HipNative_greedyCluster(JNIEnv *env, jclass c, jlong h, jint maxShift,
                                                                          jint shiftPenalty, jint threshold, jint maxClusters,
                                                                          jintArray clusterId, jintArray resultOrder,
                                                                          jintArray memberRank) {
    (void)c;
    hmk_ctx *ctx = (hmk_ctx *)(intptr_t)h;
    jint *cid = (*env)->GetIntArrayElements(env, clusterId, NULL);
    jint *ord = (*env)->GetIntArrayElements(env, resultOrder, NULL);
    jint *rank = (*env)->GetIntArrayElements(env, memberRank, NULL);
    hmk_greedy_stats stats;
    int st = hmk_greedy_cluster(ctx, maxShift, shiftPenalty, threshold, maxClusters, (int32_t *)cid, (int32_t *)ord,
                                (int32_t *)rank, &stats);
    (*env)->ReleaseIntArrayElements(env, clusterId, cid, 0);
    (*env)->ReleaseIntArrayElements(env, resultOrder, ord, 0);
    (*env)->ReleaseIntArrayElements(env, memberRank, rank, 0);
    if (st) { throw_for(env, ctx, st); return 0; }
    return stats.n_result_clusters;
}

JNIEXPORT jint JNICALL Java_cz_krejciadam_hammock_HipNative_clinkageCluster(JNIEnv *env, jclass c, jlong h, jint maxShift,
                                                                            jint shiftPenalty, jint threshold, jintArray clusterId,
                                                                            jintArray resultOrder, jintArray memberRank) {
    (void)c;
    hmk_ctx *ctx = (hmk_ctx *)(intptr_t)h;
    jint *cid = (*env)->GetIntArrayElements(env, clusterId, NULL);
    jint *ord = (*env)->GetIntArrayElements(env, resultOrder, NULL);
    jint *rank = (*env)->GetIntArrayElements(env, memberRank, NULL);
    hmk_clinkage_stats stats;
    int st = hmk_clinkage_cluster(ctx, maxShift, shiftPenalty, threshold, (int32_t *)cid, (int32_t *)ord, (int32_t *)rank, &stats);
    (*env)->ReleaseIntArrayElements(env, clusterId, cid, 0);
    (*env)->ReleaseIntArrayElements(env, resultOrder, ord, 0);
    (*env)->ReleaseIntArrayElements(env, memberRank, rank, 0);
    if (st == HMK_ERR_REFERENCE_WOULD_CRASH) {   /* activeClusters.iterator().next() on an empty set, :118 */
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/util/NoSuchElementException"), hmk_last_error(ctx));
        return 0;
    }
    if (st) { throw_for(env, ctx, st); return 0; }
    return stats.n_result_clusters;
}
