/*
 * tests/jni_stub/jni.h -- NOT the JDK's header.  A compile-check stand-in for the handful of JNI declarations
 * hammock_amd/java/jni/hammock_jni.c uses, so that the shim keeps compiling in an image without a JDK
 * (tests/test_cli_io.py::test_jni_shim_compiles).  Only this repository's own JNI glue is compiled against it, with
 * -fsyntax-only semantics in mind: the function table below has the right member TYPES, not the JVM's layout, and
 * nothing built against it may ever be loaded into a JVM.  Where a JDK exists, build with $JAVA_HOME/include instead
 * (INTEGRATION.md).
 */
#ifndef HAMMOCK_TEST_JNI_STUB_H
#define HAMMOCK_TEST_JNI_STUB_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jbyteArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    jbyte *(*GetByteArrayElements)(JNIEnv *env, jbyteArray array, jboolean *isCopy);
    void (*ReleaseByteArrayElements)(JNIEnv *env, jbyteArray array, jbyte *elems, jint mode);
    jintArray (*NewIntArray)(JNIEnv *env, jsize len);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
};

#endif
