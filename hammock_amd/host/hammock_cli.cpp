// hammock_cli.cpp -- `hammock-hip greedy ...`: the C++ counterpart of
// `java -jar Hammock.jar greedy ...` (Hammock.java:142-172, :217-234, :392-437) that drives
// the same C ABI the Java shim binds.  Flags, defaults, log lines and result files follow the
// reference's greedy mode; everything after initial clustering (Clustal Omega MSAs, HMM stage)
// is out of scope.
//
// Extra flags that the reference does not have: --device <k> (HIP ordinal, default 0) and --devices a,b,.. (several
// GPUs of the node behind one context, the first is the root: hmk_create_multi).
#include <future>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <climits>

#include <malloc.h>

#include "hammock_host.hpp"

using namespace hammock;

namespace {

const char *VERSION = "1.2.0";  // the Hammock version whose greedy mode this mirrors (Hammock.java:39)

std::string parentDir() {  // PARENT_DIR (Hammock.java:36): two levels above the binary
    char buf[PATH_MAX];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    std::string p = n > 0 ? std::string(buf, (size_t)n) : std::string("./hammock-hip");
    for (int k = 0; k < 2; k++) {
        const size_t s = p.find_last_of('/');
        p = s == std::string::npos ? std::string(".") : p.substr(0, s);
    }
    return p;
}

bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

struct Options {
    // common (Hammock.java:40-67)
    std::string inputFileName, workingDirectory, matrixFile, labelString, tempDirectory = "/tmp";
    bool haveInput = false, haveDir = false, haveLabels = false;
    int nThreads = 4;
    int seed = 42;
    // greedy (Hammock.java:79-85)
    std::string inputType = "fasta", order = "size";
    bool haveThreshold = false, haveMaxShift = false, haveLimit = false;
    int sequenceClusteringThreshold = 0, shiftPenalty = 0, maxShift = 0, initialClustersLimit = 0;
    int cacheSizeLimit = 1;   // clinkage: -L is parsed and logged, never used (Hammock.java:89,1004-1008,459)
    int device = 0;
    std::vector<int> devices;   // --devices 0,1,..: pair space sharded over several GPUs
    int javaHashSet = 8;        // --java_hashset 8|7|6 (clinkage): whose java.util.HashSet iteration order is emulated
};

void parseCommonArgs(const std::vector<std::string> &args, Options &o) {  // Hammock.java:824-908
    for (size_t i = 1; i < args.size(); i++) {
        const std::string &a = args[i];
        const bool more = args.size() > i + 1;
        if ((a == "-i" || a == "--input") && more) { o.inputFileName = args[++i]; o.haveInput = true; continue; }
        if ((a == "-d" || a == "--outputDirectory") && more) { o.workingDirectory = args[++i]; o.haveDir = true; continue; }  // :834
        if ((a == "-m" || a == "--matrix") && more) { o.matrixFile = args[++i]; continue; }
        if ((a == "-t" || a == "--threads") && more) {
            try { size_t u = 0; o.nThreads = std::stoi(args[i + 1], &u); if (u != args[i + 1].size()) throw 0; }
            catch (...) { throw HammockException("NumberFormatException: For input string: \"" + args[i + 1] + "\""); }
            i++;
            continue;
        }
        if ((a == "-l" || a == "--labels") && more) { o.labelString = args[++i]; o.haveLabels = true; continue; }
        if (a == "--temp" && more) { o.tempDirectory = args[i + 1]; }  // :903-905 (no skip, as in the reference)
        if (a == "--device" && more) { o.device = javaIntegerDecode(args[++i]); continue; }
        if (a == "--java_hashset" && more) {
            o.javaHashSet = javaIntegerDecode(args[++i]);
            if (o.javaHashSet != 8 && o.javaHashSet != 7 && o.javaHashSet != 6)
                throw CLIException("Error. --java_hashset may be 8 (Java 8 and later, the default), 7 (JDK 7u6 and later 7 updates) or 6 (JDK 6, JDK 7 before 7u6).");
            continue;
        }
        if (a == "--devices" && more) {
            o.devices.clear();
            std::string list = args[++i], tok;
            for (size_t b = 0; b <= list.size(); b++) {
                if (b == list.size() || list[b] == ',') { if (!tok.empty()) o.devices.push_back(javaIntegerDecode(tok)); tok.clear(); }
                else tok.push_back(list[b]);
            }
            if (o.devices.empty()) throw HammockException("--devices needs a comma separated list of HIP ordinals");
            continue;
        }
    }
}

void parseGreedyArgs(const std::vector<std::string> &args, Options &o) {  // Hammock.java:915-970
    for (size_t i = 1; i < args.size(); i++) {
        const std::string &a = args[i];
        const bool more = args.size() > i + 1;
        if ((a == "-f" || a == "--file_format") && more) { o.inputType = args[++i]; continue; }
        if ((a == "-g" || a == "--greedy_threshold" || a == "--alignment_threshold") && more) {
            o.sequenceClusteringThreshold = javaIntegerDecode(args[++i]); o.haveThreshold = true; continue;
        }
        if ((a == "-x" || a == "--max_shift") && more) { o.maxShift = javaIntegerDecode(args[++i]); o.haveMaxShift = true; continue; }
        if ((a == "-R" || a == "--order") && more) { o.order = args[++i]; continue; }
        if ((a == "-S" || a == "--seed") && more) { o.seed = javaIntegerDecode(args[++i]); continue; }
        if ((a == "-p" || a == "--gap_penalty") && more) { o.shiftPenalty = javaIntegerDecode(args[++i]); }
        else if (a == "--initial_clusters_limit" && more) { o.initialClustersLimit = javaIntegerDecode(args[++i]); o.haveLimit = true; }
    }
}

void parseClinkageArgs(const std::vector<std::string> &args, Options &o) {  // Hammock.java:972-1011
    for (size_t i = 1; i < args.size(); i++) {
        const std::string &a = args[i];
        const bool more = args.size() > i + 1;
        if ((a == "-f" || a == "--file_format") && more) { o.inputType = args[++i]; continue; }
        if ((a == "-x" || a == "--max_shift") && more) { o.maxShift = javaIntegerDecode(args[++i]); o.haveMaxShift = true; continue; }
        if ((a == "-p" || a == "--gap_penalty") && more) { o.shiftPenalty = javaIntegerDecode(args[++i]); }
        if ((args[i] == "-g" || args[i] == "--greedy_threshold" || args[i] == "--alignment_threshold") && args.size() > i + 1) {
            o.sequenceClusteringThreshold = javaIntegerDecode(args[++i]); o.haveThreshold = true;
        }
        if ((args[i] == "-L" || args[i] == "--cache_size_limit") && args.size() > i + 1) { o.cacheSizeLimit = javaIntegerDecode(args[++i]); }
    }
}

void printHelp() {  // Hammock.java:295-320 (greedy-relevant part)
    std::cerr << "\nhammock-hip: MI355X-native greedy and clinkage modes of Hammock version " << VERSION << "\n\n"
              << "Synopsis: hammock-hip <greedy|clinkage> <param1> <param2> ...\n\n"
              << "-i, --input <file>\n\tA path to an input file\n\n"
              << "-d, --output_directory <directory>\n\tA directory to store all output files in\n\n"
              << "-t, --threads <int>\n\tAccepted for compatibility (the GPU path ignores it)\n\n"
              << "-l, --labels <str,str,str...>\n\tA list of sequence labels to use\n\n"
              << "-f, --file_format <[fasta,tab]>\n\tThe file format of input file specified by -i\n\n"
              << "-m, --matrix <file>\n\tA path to a substitution matrix file\n\n"
              << "-g, --alignment_threshold, (--greedy_threshold) <int>\n\tMinimal score needed for a sequence to join a cluster\n\n"
              << "-x, --max_shift <int>\n\tMaximal sequence-sequence shift. A nonnegative int\n\n"
              << "-p, --gap_penalty <int>\n\tThe penalty for each position of the sequence-sequence shift. A nonpositive int\n\n"
              << "-R, --order [size, alphabetic, random, input, <label>]\n\tThe order of sequences during greedy clustering\n\n"
              << "-S, --seed <int>\n\tA seed to make random processes deterministic (if -R random is in use)\n\n"
              << "--initial_clusters_limit <int>\n\tThe max. number of clusters resulting from gredy clustering\n\n"
              << "-L, --cache_size_limit <int>\n\t(clinkage) accepted and logged; has no effect, as in the reference\n\n"
              << "--device <int>\n\tHIP device ordinal (default 0)\n\n"
              << "--devices <int,int,...>\n\tShard the pair space over several GPUs of the node (the first one runs the merge)\n\n"
              << "--java_hashset <8|7|6>\n\t(clinkage) whose java.util.HashSet iteration order picks the chain starts and orders the result: 8 = Java 8 and\n\tlater (default), 7 = JDK 7u6 and later updates of 7, 6 = JDK 6 and JDK 7 before 7u6\n\n";
}

std::string labelsToString(bool have, const std::vector<std::string> &labels) {  // List.toString() / "null"
    if (!have) return "null";
    std::string s = "[";
    for (size_t k = 0; k < labels.size(); k++) s += (k ? ", " : "") + labels[k];
    return s + "]";
}

// Hammock.java:1421-1427 (the list's shortest length comes from the one summary pass; the mean length, :1554-1563, too)
int checkMaxShift(const SequenceListSummary &summary, int maxShift) { return std::min(maxShift, summary.minLength - 1); }

// greedy mode (Hammock.java:217-234, runGreedyClustering :392-437) and clinkage mode (:236-253, runClinkageClustering
// :449-489): the two share everything but the clusterer, the ordering step and a few log lines
int runSequenceClustering(const std::vector<std::string> &args, bool clinkage) {
    Options o;
    const std::string PARENT_DIR = parentDir();
    o.matrixFile = PARENT_DIR + "/matrices/blosum62.txt";  // Hammock.java:45
    parseCommonArgs(args, o);
    if (clinkage) parseClinkageArgs(args, o);
    else parseGreedyArgs(args, o);

    // ---- checkCommonArgs, Hammock.java:1207-1266 ------------------------------------------------
    if (!o.haveInput) throw CLIException("Error. Parameter input file (-i or --input) missing with no default.");
    if (o.haveDir) {
        if (exists(o.workingDirectory)) throw CLIException("Error. Output directory exists. Exiting to prevent data loss.");
        mkdir(o.workingDirectory.c_str(), 0777);
    } else {
        std::string name;
        mkdir((PARENT_DIR + "/dist").c_str(), 0777);
        for (int i = 1; i < 9999; i++) {
            name = PARENT_DIR + "/dist/Hammock_result_" + std::to_string(i);
            if (!exists(name)) { mkdir(name.c_str(), 0777); break; }
        }
        o.workingDirectory = name;
        std::cerr << "Creating default output directory: " << name << std::endl;
    }
    Logger logger(o.workingDirectory + "/run.log", false);
    try {
        logger.logAndStderr(std::string("\nHammock version ") + VERSION +
                            " Run with --help for a brief description of command line parameters.\n");
        std::vector<std::string> labels;
        if (o.haveLabels) labels = FileIOManager::splitChar(o.labelString, ',', true);
        const std::string initialClustersSequencesCsv = o.workingDirectory + "/initial_clusters_sequences.tsv";
        const std::string initialClustersSequencesOrderedCsv = o.workingDirectory + "/initial_clusters_sequences_original_order.tsv";
        const std::string initialClusters = o.workingDirectory + "/initial_clusters.tsv";
        const std::string inputStatistics = o.workingDirectory + "/input_statistics.tsv";
        const std::vector<std::vector<int>> scoringMatrix = FileIOManager::loadScoringMatrix(o.matrixFile);  // :1264
        // the GPU context (HIP start-up, queues, code objects: 70-150 ms) is created while the input is read and summarised
        std::shared_future<std::shared_ptr<NativeContext>> contextReady = std::async(std::launch::async, [&scoringMatrix, &o]() {
            std::shared_ptr<NativeContext> c = o.devices.empty() ? std::make_shared<NativeContext>(scoringMatrix, o.device)
                                                                 : std::make_shared<NativeContext>(scoringMatrix, o.devices);
            if (o.javaHashSet != 8 && hmk_set_java_hashset(c->get(), o.javaHashSet) != HMK_OK) throw HammockException("hmk_set_java_hashset failed");
            return c;
        });
        // ---- checkGreedyOrClinkageArgs, :1272-1277 ------------------------------------------------
        if (!(o.inputType == "fasta" || o.inputType == "seq" || o.inputType == "tab"))
            throw CLIException("Error. Parameter -f value may be either \"fasta\", \"seq\" or \"tab\". No other values are allowed");

        logger.logWithTime(clinkage ? "Program started in mode \"clinkage\"." : "Program started in mode \"greedy\".");  // :225-229, :244-248
        std::string argsString;
        for (auto &a : args) argsString += " " + a;
        logger.logWithoutTime("Command-line arguments: \n" + argsString + "\n");
        logger.logWithoutTime("\nComplete list of input/output parameters: \n-i, --input " + o.inputFileName +
                              "\n-d, --output_directory " + o.workingDirectory + "\n-t, --thread " + std::to_string(o.nThreads) +
                              "\n-l, --labels " + labelsToString(o.haveLabels, labels) + "\n\n");
        if (clinkage)   // logClinkageParams, :1742-1755 (labels as the reference prints them)
            logger.logWithoutTime("\nComplete list of clinkage clustering parameters: \n-f, --file_format " + o.inputType +
                                  "\n-m, --matrix " + o.matrixFile + "\n-g, --alignment_threshold (--greedy_threshold)" +
                                  (o.haveThreshold ? std::to_string(o.sequenceClusteringThreshold) : std::string("null")) +
                                  "\n-x, --max_shift " + (o.haveMaxShift ? std::to_string(o.maxShift) : std::string("null")) +
                                  "\n-p, --gap_penalty " + std::to_string(o.shiftPenalty) + "\n-C, --cache_size_limit " +
                                  std::to_string(o.cacheSizeLimit) + "\n\n");
        else
        logger.logWithoutTime("\nComplete list of greedy clustering parameters: \n-f, --file_format " + o.inputType +
                              "\n-m, --matrix " + o.matrixFile + "\n-g, --greedy_threshold " +
                              (o.haveThreshold ? std::to_string(o.sequenceClusteringThreshold) : std::string("null")) +
                              "\n-x, --max_shift " + (o.haveMaxShift ? std::to_string(o.maxShift) : std::string("null")) +
                              "\n-p, --gap_penalty " + std::to_string(o.shiftPenalty) + "\n-R, --order " + o.order +
                              "\n-S, --seed " + std::to_string(o.seed) + "\n\n");

        // ---- loadInputSequences, :749-787 -----------------------------------------------------------
        const auto timeStart = std::chrono::steady_clock::now();
        auto cliLap = [&](const char *what) {   // HMK_CLI_TIMING=1: where a hammock-hip process spends its time
            if (std::getenv("HMK_CLI_TIMING"))
                std::fprintf(stderr, "[hammock-hip] %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - timeStart).count());
        };
        logger.logAndStderr("Loading input sequences...");
        std::vector<UniqueSequencePtr> sequences;
        if (o.inputType == "fasta") sequences = FileIOManager::loadUniqueSequencesFromFasta(o.inputFileName);
        else if (o.inputType == "tab") sequences = FileIOManager::loadUniqueSequencesFromTable(o.inputFileName);
        else throw HammockException("Error, this should have been checked.");  // "seq", :759-761
        cliLap("input loaded");
        logger.logAndStderr(std::to_string(sequences.size()) + " unique sequences loaded.");
        SequenceListSummary summary = summariseSequences(sequences);
        logger.logAndStderr(std::to_string(summary.total) + " total sequences loaded.");
        if (o.haveLabels) {  // filterSequencesForLabels, :1661-1675
            std::vector<UniqueSequencePtr> kept;
            for (auto &s : sequences) {
                std::vector<std::pair<std::string, int>> lm;
                for (auto &label : labels) {
                    bool present = false;
                    const int c = s->labelCount(label, &present);
                    if (present) lm.push_back({label, c});
                }
                if (!lm.empty()) kept.push_back(std::make_shared<UniqueSequence>(s->getSequenceString(), lm));
            }
            sequences = kept;
            summary = summariseSequences(sequences);
        }
        logger.logAndStderr(std::to_string(sequences.size()) + " unique sequences after non-specified labels filtered out");
        logger.logAndStderr(std::to_string(summary.total) + " total sequences after non-specified labels fileterd out");
        const int minLength = summary.minLength, maxLength = summary.maxLength;
        logger.logAndStderr("Shortest sequence: " + std::to_string(minLength) + " AA. Longest sequence: " + std::to_string(maxLength) + " AA.");
        if (sequences.empty()) throw FileFormatException("Error. No sequences (with specified labels) to cluster.");
        // the sequence count is known: the context's buffers (24 GB at 10^6) are sized on another thread while this one goes on
        // to the labels, the statistics, the sort and the upload
        std::future<void> reserved = std::async(std::launch::async, [contextReady, timeStart, count = (uint32_t)sequences.size()]() {
            try {   // (a device error is reported by the clustering call)
                hmk_ctx *c = contextReady.get()->get();
                if (std::getenv("HMK_CLI_TIMING")) std::fprintf(stderr, "[hammock-hip] hmk_reserve begins at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - timeStart).count());
                (void)hmk_reserve(c, count);
                if (std::getenv("HMK_CLI_TIMING")) std::fprintf(stderr, "[hammock-hip] hmk_reserve done at %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - timeStart).count());
            } catch (...) { }
        });
        if (maxLength > HMK_MAX_LEN)   // the reference has no such limit; say so here instead of failing inside the clusterer
            throw HammockException("Error. The longest sequence has " + std::to_string(maxLength) + " amino acids; the GPU kernels of hammock-hip "
                                   "take sequences of up to " + std::to_string(HMK_MAX_LEN) + " (Hammock's domain is 7-20).");

        // ---- runGreedyClustering, :392-437 ------------------------------------------------------------
        if (!o.haveLabels) labels = FileIOManager::getSortedLabels(sequences);                 // :796-798
        const std::vector<UniqueSequencePtr> initialSequences(sequences);                       // :800-801
        if (!o.haveMaxShift) {                                                                  // :803-811
            o.maxShift = checkMaxShift(summary, (int)javaRound(summary.meanLength() / 4));
            logger.logAndStderr("Max shift not set. Setting automatically to: " + std::to_string(o.maxShift));
        } else {
            const int correct = checkMaxShift(summary, o.maxShift);
            if (o.maxShift != correct) {
                o.maxShift = correct;
                logger.logAndStderr("Setting max shift to " + std::to_string(correct) +
                                    " as the length of the shortest sequence is only " + std::to_string(correct + 1));
            }
        }
        cliLap("labels, lengths, max shift");
        logger.logAndStderr("Generating input statistics...");
        FileIOManager::saveInputStatistics(sequences, labels, inputStatistics);                 // :814-816
        cliLap("input statistics written");
        if (!o.haveThreshold) {                                                                 // :394-397 / :452-455
            o.sequenceClusteringThreshold = (int)javaRound(summary.meanLength() * 1.7);
            logger.logAndStderr(std::string(clinkage ? "Clinkage" : "Greedy") + " clustering threshold not set. Setting automatically to: " +
                                std::to_string(o.sequenceClusteringThreshold));
        }
        if (!clinkage && !o.haveLimit) {                                                                     // :398-401
            o.initialClustersLimit = (int)javaRound((double)sequences.size() * 0.025);
            logger.logAndStderr("Initial greedy clusters limit not set. Setting automatically to: " +
                                std::to_string(o.initialClustersLimit));
        }
        auto scorer = std::make_shared<ShiftedScorer>(contextReady.get(), o.shiftPenalty, o.maxShift);  // :402 (get() rethrows a device error)
        HipGreedySequenceClusterer clusterer(scorer, o.sequenceClusteringThreshold, o.initialClustersLimit);  // :403
        HipClinkageSequenceClusterer clinkageClusterer(scorer, o.sequenceClusteringThreshold);                // :459

        cliLap("GPU context ready");
        logger.logAndStderr(clinkage ? "Clinkage clustering..." : "Greedy clustering...");
        const auto time0 = std::chrono::steady_clock::now();
        if (!clinkage) sortSequences(sequences, o.order, o.seed, labels);                       // :407 (clinkage keeps the load order)
        if (std::getenv("HMK_CLI_TIMING"))
            std::fprintf(stderr, "[hammock-hip] sort: %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - time0).count());
        std::vector<ClusterPtr> clusters = clinkage ? clinkageClusterer.cluster(sequences)      // :462
                                                    : clusterer.cluster(sequences);             // :409
        auto ms = [&]() {
            return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - time0).count();
        };
        cliLap("clustered");
        logger.logAndStderr("Ready. Clustering time: " + std::to_string(ms()));                // :411 / :463
        logger.logAndStderr("Resulting clusers: " + std::to_string(clusters.size()));          // :412 / :464
        if (clinkage)
            logger.logAndStderr("GPU scoring: " + std::to_string(clinkageClusterer.stats.neighbors_ms) + " ms, host nearest-neighbour chain: " +
                                std::to_string(clinkageClusterer.stats.chain_ms) + " ms, neighbour edges: " +
                                std::to_string(clinkageClusterer.stats.n_edges) + ", merges: " + std::to_string(clinkageClusterer.stats.merges));
        else
        logger.logAndStderr("GPU scoring + adjacency build: " + std::to_string(clusterer.stats.neighbors_ms) + " ms, host greedy merge: " +
                            std::to_string(clusterer.stats.greedy_ms) + " ms, neighbour edges: " +
                            std::to_string(clusterer.stats.n_edges));
        logger.logAndStderr("Building MSAs... (skipped: Clustal Omega is outside the scope of hammock-hip; the alignment "
                            "column of multi-member clusters is NA)");
        logger.logAndStderr("Ready. Total time: " + std::to_string(ms()));                     // :427
        logger.logAndStderr("Saving results to output files...");
        // the reference's three calls in a row (:429, :431, :432), written side by side
        FileIOManager::saveInitialClusters(clusters, initialClustersSequencesCsv, initialClustersSequencesOrderedCsv, initialClusters,
                                           labels, initialSequences);
        cliLap("result files written");
        logger.logAndStderr(std::string(clinkage ? "Clinkage" : "Greedy") + " clustering results in: " + initialClusters);
        logger.logAndStderr("and: " + initialClustersSequencesCsv);
        logger.logAndStderr("and: " + initialClustersSequencesOrderedCsv);
        logger.logWithTime("Program successfully ended.");
        // Everything is on disk (the writers and the logger close their files).  Tearing down 10^6 sequence and cluster
        // objects one by one and handing 36 GB of device buffers back costs 0.3-0.5 s that change nothing: leave at once,
        // the driver reclaims the device memory with the process.  (Handing the
        // context back on another thread while the files are written was measured too: the writers lose 0.1 s to it and the
        // process still needs 0.17 s to go, 1.42-1.51 s against 1.17-1.45 s.)
        cliLap("log closed, leaving");
        std::cout.flush();
        std::cerr.flush();
        std::fflush(nullptr);
        std::_Exit(0);
        return 0;
    } catch (const CLIException &) {
        throw;
    } catch (const FileFormatException &e) {  // Hammock.java:148-152
        logger.logAndStderr("Error. Probably wrong input file format? Run with --help for a brief description of command line parameters. Trace: \n");
        logger.logAndStderr(std::string("cz.krejciadam.hammock.FileFormatException: ") + e.what());
        return 3;
    } catch (const NullPointerException &e) {  // :153-157
        logger.logAndStderr("Error. Maybe wrong input file format? Run with --help for a brief description of command line parameters. Trace: \n");
        logger.logAndStderr(std::string("java.lang.NullPointerException: ") + e.what());
        return 4;
    } catch (const DataException &e) {  // :158-162
        logger.logAndStderr("Error. Maybe wrong input file format or wrong set of labels? Run with --help for a brief description of command line parameters. Trace: \n");
        logger.logAndStderr(std::string("cz.krejciadam.hammock.DataException: ") + e.what());
        return 5;
    } catch (const std::exception &e) {  // :163-167
        logger.logAndStderr("Error. Run with --help for a brief description of command line parameters. Trace: \n");
        logger.logAndStderr(e.what());
        return 6;
    }
}

// `hammock-hip io-selftest ...`: exposes the loaders / orderings to the CPU test-suite (no GPU involved)
int ioSelftest(const std::vector<std::string> &args) {
    if (args.size() >= 3 && args[1] == "matrix") {
        for (auto &row : FileIOManager::loadScoringMatrix(args[2])) {
            for (size_t c = 0; c < row.size(); c++) std::cout << (c ? " " : "") << row[c];
            std::cout << "\n";
        }
        return 0;
    }
    if (args.size() >= 5 && args[1] == "sequences") {  // sequences <fasta|tab> <file> <order> [seed]
        auto seqs = args[2] == "tab" ? FileIOManager::loadUniqueSequencesFromTable(args[3])
                                     : FileIOManager::loadUniqueSequencesFromFasta(args[3]);
        const std::vector<std::string> labels = FileIOManager::getSortedLabels(seqs);
        sortSequences(seqs, args[4], args.size() > 5 ? javaIntegerDecode(args[5]) : 42, labels);
        std::cout << "labels";
        for (auto &l : labels) std::cout << "\t" << l;
        std::cout << "\n";
        for (auto &s : seqs) std::cout << s->getSequenceString() << "\t" << FileIOManager::sequenceLine(*s, labels) << "\n";
        return 0;
    }
    if (args.size() >= 8 && args[1] == "writers") {  // writers <fasta|tab> <file> <order> <seed> <clusters.tsv> <out dir>
        // clusters.tsv: one cluster per line, "id<TAB>SEQ,SEQ,..." in list order; the result files are written twice, by the
        // reference's three calls in a row (<out dir>/serial) and side by side (<out dir>/side)
        auto seqs = args[2] == "tab" ? FileIOManager::loadUniqueSequencesFromTable(args[3])
                                     : FileIOManager::loadUniqueSequencesFromFasta(args[3]);
        const std::vector<std::string> labels = FileIOManager::getSortedLabels(seqs);
        const std::vector<UniqueSequencePtr> initial(seqs);
        FileIOManager::saveInputStatistics(seqs, labels, args[7] + "/input_statistics.tsv");
        sortSequences(seqs, args[4], javaIntegerDecode(args[5]), labels);
        std::unordered_map<std::string, UniqueSequencePtr> byString;
        for (auto &q : seqs) byString[q->getSequenceString()] = q;
        auto build = [&]() {
            std::vector<ClusterPtr> clusters;
            for (const std::string &line : FileIOManager::readLines(args[6])) {
                const std::vector<std::string> f = FileIOManager::splitChar(line, '\t', true);
                if (f.size() < 2) continue;
                std::vector<UniqueSequencePtr> members;
                for (const std::string &m : FileIOManager::splitChar(f[1], ',', true)) members.push_back(byString.at(m));
                clusters.push_back(std::make_shared<Cluster>(members, javaIntegerDecode(f[0])));
            }
            return clusters;
        };
        std::vector<ClusterPtr> a = build(), b = build();
        FileIOManager::saveClusterSequencesToCsv(a, args[7] + "/serial/initial_clusters_sequences.tsv", labels);
        FileIOManager::saveClusterSequencesToCsvOrdered(a, args[7] + "/serial/initial_clusters_sequences_original_order.tsv", labels, initial);
        FileIOManager::SaveClustersToCsv(a, args[7] + "/serial/initial_clusters.tsv", labels);
        FileIOManager::saveInitialClusters(b, args[7] + "/side/initial_clusters_sequences.tsv", args[7] + "/side/initial_clusters_sequences_original_order.tsv",
                                           args[7] + "/side/initial_clusters.tsv", labels, initial);
        return 0;
    }
    std::cerr << "usage: hammock-hip io-selftest matrix <file> | sequences <fasta|tab> <file> <order> [seed] | "
                 "writers <fasta|tab> <file> <order> <seed> <clusters.tsv> <out dir>\n";
    return 2;
}

// `hammock-hip api-selftest <known_answers.tsv> <matrix>`: drives the mirrored C++ classes
// (ShiftedScorer, LocalAlignmentScorer, Cluster, HipGreedySequenceClusterer) the way a unit test of the
// reference would -- one sequenceScore / cluster call at a time -- and prints the results.
// Lines: "shifted seq1 seq2 X p" | "local seq1 seq2 open ext" | "greedy thr maxShift penalty maxClusters seq..."
int apiSelftest(const std::vector<std::string> &args) {
    if (args.size() < 3) { std::cerr << "usage: hammock-hip api-selftest <cases.tsv> <matrix file> [device]\n"; return 2; }
    const auto M = FileIOManager::loadScoringMatrix(args[2]);
    const int device = args.size() > 3 ? javaIntegerDecode(args[3]) : 0;
    for (const std::string &line : FileIOManager::readLines(args[1])) {
        const std::vector<std::string> f = FileIOManager::splitChar(line, '\t', true);
        if (f.empty() || f[0].empty() || f[0][0] == '#') continue;
        try {
            if (f[0] == "shifted" && f.size() >= 5) {
                ShiftedScorer sc(M, javaIntegerDecode(f[4]), javaIntegerDecode(f[3]), device);
                const AligningScorerResult r = sc.scoreWithShift(std::make_shared<UniqueSequence>(f[1]),
                                                                 std::make_shared<UniqueSequence>(f[2]));
                std::cout << "shifted\t" << f[1] << "\t" << f[2] << "\t" << r.getScore() << "\t" << r.getShift() << "\n";
            } else if (f[0] == "local" && f.size() >= 5) {
                LocalAlignmentScorer sc(M, javaIntegerDecode(f[3]), javaIntegerDecode(f[4]), device);
                std::cout << "local\t" << f[1] << "\t" << f[2] << "\t"
                          << sc.sequenceScore(std::make_shared<UniqueSequence>(f[1]), std::make_shared<UniqueSequence>(f[2])) << "\n";
            } else if (f[0] == "greedy" && f.size() >= 6) {
                auto scorer = std::make_shared<ShiftedScorer>(M, javaIntegerDecode(f[3]), javaIntegerDecode(f[2]), device);
                HipGreedySequenceClusterer clusterer(scorer, javaIntegerDecode(f[1]), javaIntegerDecode(f[4]));
                std::vector<UniqueSequencePtr> seqs;
                for (size_t k = 5; k < f.size(); k++) seqs.push_back(std::make_shared<UniqueSequence>(f[k]));
                std::ostringstream os;
                os << "greedy";
                for (auto &cl : clusterer.cluster(seqs)) {
                    os << "\t" << cl->getId() << ":";
                    for (size_t k = 0; k < cl->getSequences().size(); k++)
                        os << (k ? "," : "") << cl->getSequences()[k]->getSequenceString();
                }
                std::cout << os.str() << "\n";
            }
        } catch (const DataException &e) {
            std::cout << f[0] << "\tDataException\t" << e.what() << "\n";
        } catch (const NullPointerException &e) {
            std::cout << f[0] << "\tNullPointerException\tcase " << e.crashCase << " index " << e.crashIndex << "\n";
        }
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    // 10^6 sequences are 4 x 10^6 small allocations made on 16 threads: glibc grows a thread's arena by `top_pad` (128 KB) per
    // mprotect call, and every such call stops the page faults of all other threads; larger steps, fewer calls
    mallopt(M_TOP_PAD, 64 << 20);
    std::vector<std::string> args(argv + 1, argv + argc);
    if (args.empty() || args[0] == "--help" || args[0] == "-h") { printHelp(); return args.empty() ? 2 : 0; }
    try {
        if (args[0] == "greedy") return runSequenceClustering(args, false);
        if (args[0] == "clinkage") return runSequenceClustering(args, true);
        if (args[0] == "io-selftest") return ioSelftest(args);
        if (args[0] == "dump-matrix") {   // the default matrix in the text format FileIOManager.loadScoringMatrix reads
            std::cout << "# BLOSUM62 substitution matrix (public NCBI table), 24 x 24, order " << AMINO_ACIDS << "\n"
                      << "# default of hammock-hip greedy (-m), same text format Hammock's -m files use\n  ";
            for (int c = 0; c < 24; c++) std::cout << "  " << AMINO_ACIDS[c];
            std::cout << "\n";
            for (int r = 0; r < 24; r++) {
                std::cout << AMINO_ACIDS[r];
                for (int c = 0; c < 24; c++) {
                    const std::string v = std::to_string(BLOSUM62[r][c]);
                    std::cout << std::string(3 - v.size(), ' ') << v;
                }
                std::cout << "\n";
            }
            return 0;
        }
        if (args[0] == "api-selftest") return apiSelftest(args);
        std::cerr << "hammock-hip implements Hammock's initial-clustering modes `greedy` and `clinkage` (modes full, cluster, "
                     "compare drive external HMM tools and are outside the scope of the MI355X hot path); got mode \"" << args[0] << "\"\n";
        return 2;
    } catch (const CLIException &e) {  // Hammock.java:146-147
        std::cerr << "Error in command line arguments: " << e.what() << std::endl;
        return 2;
    } catch (const FileFormatException &e) {
        std::cerr << "Error. Probably wrong input file format? Run with --help for a brief description of command line parameters. Trace: \n"
                  << e.what() << std::endl;
        return 3;
    } catch (const std::exception &e) {
        std::cerr << "Error. Run with --help for a brief description of command line parameters. Trace: \n" << e.what() << std::endl;
        return 6;
    }
}
