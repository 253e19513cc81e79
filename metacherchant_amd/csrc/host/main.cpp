// metacherchant -- native launcher of the MI355X environment-finder path.  Mirrors the command line
// of the reference (src/Runner.java + src/tools/EnvironmentFinderMain.java:32-105 + the launch
// options of itmo!/utils/tool/Tool.java:59-143) and its output surface, and
// drives libmcgpu.so through the C ABI of include/mcgpu.h exactly where the Java tool calls
// IOUtils.loadReads / LargeKIOUtils.loadReads and OneSequenceCalculator.runBfs.
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "envfinder.h"
#include "mcgpu.h"

using namespace mch;

namespace {

FILE *g_log = nullptr, *g_log2 = nullptr;
bool g_verbose = false;

void logline(const char *level, const std::string &msg)
{
    char ts[32];
    const time_t now = time(nullptr);
    strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", localtime(&now));
    const bool debug = strcmp(level, "DEBUG") == 0;
    if (!debug || g_verbose) fprintf(stderr, "%s %s: %s\n", ts, level, msg.c_str());
    for (FILE *f : {g_log, g_log2})
        if (f) { fprintf(f, "%s %s: %s\n", ts, level, msg.c_str()); fflush(f); }
}
void info(const std::string &m) { logline("INFO", m); }

std::string group_digits(unsigned long long v)  // itmo!/utils/NumUtils.java:163-174
{
    std::string vs = std::to_string(v), ans;
    while (vs.size() > 3) {
        ans = "'" + vs.substr(vs.size() - 3) + ans;
        vs = vs.substr(0, vs.size() - 3);
    }
    return vs + ans;
}

std::string shorten_label(const std::string &label, int k)  // src/utils/StringUtils.java:43-49
{
    if ((int)label.size() >= 2 * k)
        return label.substr(0, (size_t)k) + "..." + label.substr(label.size() - (size_t)k) + " (length=" +
               std::to_string(label.size()) + ")";
    return label;
}

struct Options {
    int k = -1;
    std::vector<std::string> reads;
    std::string seq, hicseq, output, output_dir, work_dir = "workDir", hash = "poly", tool = "environment-finder";
    long long maxkmers = -1, maxradius = -1;
    int coverage = 1, chunklength = 1, device = 0, geneid = 1;
    std::vector<std::string> env;
    bool bothdirs = false, forcehash = false, trim = false, merge = false, cont = false, force = false, help = false;
    unsigned long long capacity_hint = 0;
    std::vector<int> devices;  // --devices 0,1,...: several GPUs as one table (mc_group_*); empty: --device alone
};

struct OptSpec { const char *name; const char *shortopt; int kind; };  // kind: 0 value, 1 bool (optional arg), 2 multi
const OptSpec SPECS[] = {
    {"k", "k", 0}, {"reads", "i", 2}, {"seq", nullptr, 0}, {"hicseq", nullptr, 0}, {"output", "o", 0},
    {"maxkmers", nullptr, 0}, {"maxradius", nullptr, 0}, {"coverage", nullptr, 0}, {"bothdirs", nullptr, 1},
    {"chunklength", nullptr, 0}, {"forcehash", nullptr, 1}, {"hash", nullptr, 0}, {"trim", nullptr, 1},
    {"merge", nullptr, 1}, {"work-dir", "w", 0}, {"available-processors", "p", 0}, {"memory", "m", 0},
    {"continue", "c", 1}, {"force", nullptr, 1}, {"verbose", "v", 1}, {"help", "h", 1}, {"tool", "t", 0},
    {"device", nullptr, 0}, {"devices", nullptr, 0}, {"capacity-hint", nullptr, 0}, {"output-dir", nullptr, 0}, {"env", "e", 2}, {"geneid", "g", 0},
};

const OptSpec *find_spec(const std::string &tok)
{
    for (const OptSpec &s : SPECS) {
        if (tok == std::string("--") + s.name) return &s;
        if (s.shortopt && tok == std::string("-") + s.shortopt) return &s;
    }
    return nullptr;
}

bool java_bool(const std::string &s)  // new Boolean(String): true iff equalsIgnoreCase("true")
{
    return s.size() == 4 && tolower(s[0]) == 't' && tolower(s[1]) == 'r' && tolower(s[2]) == 'u' && tolower(s[3]) == 'e';
}

long long parse_int(const std::string &name, const std::string &v)
{
    char *end = nullptr;
    errno = 0;
    const long long x = strtoll(v.c_str(), &end, 10);
    if (errno || end == v.c_str() || *end) throw Error("Can't convert value '" + v + "' of parameter '" + name + "' to type 'Integer'");
    return x;
}

Options parse_args(int argc, char **argv)
{
    std::map<std::string, std::vector<std::string>> got;
    for (int i = 1; i < argc; i++) {
        std::string tok = argv[i], inline_val;
        bool has_inline = false;
        const size_t eq = tok.find('=');
        if (tok.rfind("--", 0) == 0 && eq != std::string::npos) {  // --opt=value
            inline_val = tok.substr(eq + 1);
            tok = tok.substr(0, eq);
            has_inline = true;
        }
        const OptSpec *s = find_spec(tok);
        if (!s) throw Error("Cannot parse command line: Unrecognized option: " + tok);
        auto &vals = got[s->name];
        vals.clear();
        if (has_inline) {
            vals.push_back(inline_val);
        } else if (s->kind == 2) {
            while (i + 1 < argc && !find_spec(argv[i + 1]) && argv[i + 1][0] != '-') vals.push_back(argv[++i]);
        } else if (s->kind == 1) {
            if (i + 1 < argc && argv[i + 1][0] != '-') vals.push_back(argv[++i]);
            else vals.push_back("true");  // option without an argument, itmo!/utils/tool/Tool.java:650-652
        } else {
            if (i + 1 >= argc) throw Error("Cannot parse command line: Missing argument for option: " + tok);
            vals.push_back(argv[++i]);
        }
    }
    Options o;
    auto val = [&](const char *n) -> const std::string * { auto it = got.find(n); return it == got.end() || it->second.empty() ? nullptr : &it->second[0]; };
    if (auto v = val("k")) o.k = (int)parse_int("k", *v);
    auto multi = [&](const char *name, std::vector<std::string> &dst) {
        if (!got.count(name)) return;
        for (const std::string &v : got[name]) {  // arrays are re-tokenised on "[, ]" (Tool.java:888-895)
            std::string cur;
            for (char c : v + " ") {
                if (c == '[' || c == ',' || c == ' ' || c == ']') { if (!cur.empty()) dst.push_back(cur); cur.clear(); }
                else cur.push_back(c);
            }
        }
    };
    multi("reads", o.reads);
    multi("env", o.env);
    if (auto v = val("seq")) o.seq = *v;
    if (auto v = val("hicseq")) o.hicseq = *v;
    if (auto v = val("output")) o.output = *v;
    if (auto v = val("maxkmers")) o.maxkmers = parse_int("maxkmers", *v);
    if (auto v = val("maxradius")) o.maxradius = parse_int("maxradius", *v);
    if (auto v = val("coverage")) o.coverage = (int)parse_int("coverage", *v);
    if (auto v = val("chunklength")) o.chunklength = (int)parse_int("chunklength", *v);
    if (auto v = val("bothdirs")) o.bothdirs = java_bool(*v);
    if (auto v = val("forcehash")) o.forcehash = java_bool(*v);
    if (auto v = val("trim")) o.trim = java_bool(*v);
    if (auto v = val("merge")) o.merge = java_bool(*v);
    if (auto v = val("continue")) o.cont = java_bool(*v);
    if (auto v = val("force")) o.force = java_bool(*v);
    if (auto v = val("verbose")) g_verbose = java_bool(*v);
    if (auto v = val("help")) o.help = java_bool(*v);
    if (auto v = val("hash")) o.hash = *v;
    if (auto v = val("work-dir")) o.work_dir = *v;
    if (auto v = val("tool")) o.tool = *v;
    if (auto v = val("output-dir")) o.output_dir = *v;
    if (auto v = val("geneid")) o.geneid = (int)parse_int("geneid", *v);
    if (auto v = val("device")) o.device = (int)parse_int("device", *v);
    if (auto v = val("capacity-hint")) o.capacity_hint = (unsigned long long)parse_int("capacity-hint", *v);
    if (auto v = val("devices")) {  // "0,1,2" or "0-7"
        const std::string &d = *v;
        const size_t dash = d.find('-');
        if (dash != std::string::npos && d.find(',') == std::string::npos) {
            const long long a = parse_int("devices", d.substr(0, dash)), b = parse_int("devices", d.substr(dash + 1));
            for (long long i = a; i <= b; i++) o.devices.push_back((int)i);
        } else {
            size_t at = 0;
            while (at <= d.size()) {
                const size_t comma = d.find(',', at);
                const std::string tok = d.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
                if (!tok.empty()) o.devices.push_back((int)parse_int("devices", tok));
                if (comma == std::string::npos) break;
                at = comma + 1;
            }
        }
        if (o.devices.empty() || o.devices.size() > 64) throw Error("--devices: give 1 to 64 GPU ordinals, e.g. 0,1,2,3 or 0-7");
    }
    return o;
}

void usage()
{
    puts("MetaCherchant: genomic environment analysis tool (MI355X-native environment-finder path)\n");
    puts("Usage:     metacherchant [<Launch options>] [<Input parameters>]\n");
    puts("Input parameters of --tool environment-finder:");
    puts("  -k, --k <arg>            k-mer size (MANDATORY)");
    puts("  -i, --reads <args>       FASTQ, FASTA reads");
    puts("      --seq <arg>          FASTA file with sequences (MANDATORY)");
    puts("      --hicseq <arg>       FASTA file with Hi-C sequences");
    puts("  -o, --output <arg>       output directory (MANDATORY)");
    puts("      --maxkmers <arg>     maximum number of k-mers in created subgraph");
    puts("      --maxradius <arg>    maximum distance in k-mers from starting gene");
    puts("      --coverage <arg>     minimum depth of k-mers to consider (default 1)");
    puts("      --bothdirs [<arg>]   run graph search in both directions from starting sequence (default false)");
    puts("      --chunklength <arg>  minimum node length for BLAST search (default 1)");
    puts("      --forcehash [<arg>]  force k-mer hashing (even for k <= 31) (default false)");
    puts("      --hash <arg>         hash function to use: poly or fnv1a (default poly)");
    puts("      --trim [<arg>]       trim all not maximal paths? (default false)");
    puts("      --merge [<arg>]      draw single environment for multiple input sequences? (default false)");
    puts("Input parameters of --tool environment-finder-multi (CPU only): -e/--env <graph.txt files>, --seq, -o/--output, -g/--geneid (default 1)");
    puts("Input parameters of --tool kmer-counter: -k, -i/--reads, --hash, --output-dir <dir> (default <work-dir>/kmers)");
    puts("Launch options: -w/--work-dir <dir> (default workDir), -c/--continue, --force, -v/--verbose, -h/--help,");
    puts("                -t/--tool <name>, -p/--available-processors <n> and -m/--memory <arg> (accepted, unused),");
    puts("                --device <n> (GPU ordinal), --devices <a,b,...|a-b> (several GPUs as one table: reads dealt to them,");
    puts("                k-mers exchanged by owner over xGMI, BFS on the first), --capacity-hint <distinct k-mers>");
    puts("                (sizes the table once; with -k 33..63 every batch then travels as super-k-mer records, not only the first)");
}

#define MC_CHECK(ctx, call)                                                       \
    do {                                                                          \
        const int rc_ = (call);                                                   \
        if (rc_ != MC_OK) throw Error(std::string(mc_last_error(ctx)));           \
    } while (0)

struct CtxGuard {
    mc_ctx *c = nullptr;
    ~CtxGuard() { mc_destroy(c); }
};

// buildEnvironment (src/algo/OneSequenceCalculator.java:137-144): the passes of one calculator
std::vector<int> pass_dirs(bool bothdirs) { return bothdirs ? std::vector<int>{0} : std::vector<int>{-1, 1}; }

// work dir: log files, in.properties / SUCCESS (itmo!/utils/tool/Tool.java:31-33,318-392,666-689).
// Returns false when --continue finds the tool finished already.
bool open_work_dir(const Options &o, const std::string &props)
{
    const std::string wd = o.work_dir;
    write_file(wd + "/logs/.keep", "");
    char ts[32];
    const time_t now = time(nullptr);
    strftime(ts, sizeof ts, "%Y.%m.%d_%H.%M.%S", localtime(&now));
    g_log = fopen((wd + "/log").c_str(), "w");
    g_log2 = fopen((wd + "/logs/log_" + ts).c_str(), "w");
    struct stat st;
    if (stat((wd + "/in.properties").c_str(), &st) == 0 && !o.force && !o.cont)
        logline("WARN", "Work directory " + wd + " holds a previous run; overwriting (the reference would prompt; pass --force to silence)");
    if (o.cont && stat((wd + "/SUCCESS").c_str(), &st) == 0) {
        info("Tool " + o.tool + " already finished in " + wd + " (--continue), nothing to do");
        return false;
    }
    remove((wd + "/SUCCESS").c_str());
    write_file(wd + "/in.properties", props);
    return true;
}

// One table on one GPU (mc_*), or on several (mc_group_*: --devices): the calls environment-finder makes
struct Engine {
    mc_ctx *c = nullptr;
    mc_group *g = nullptr;
    ~Engine() { if (g) mc_group_destroy(g); else mc_destroy(c); }
    void open(const mc_config &cfg, const std::vector<int> &devices)
    {
        if (devices.empty()) {
            if (mc_create(&cfg, &c) != MC_OK) throw Error(std::string(mc_last_error(nullptr)));
        } else {
            std::vector<int32_t> d(devices.begin(), devices.end());
            if (mc_group_create(&cfg, d.data(), (uint32_t)d.size(), &g) != MC_OK) throw Error(std::string(mc_group_last_error(nullptr)));
        }
    }
    void check(int rc) const { if (rc != MC_OK) throw Error(std::string(g ? mc_group_last_error(g) : mc_last_error(c))); }
    void set_coverage_hint(int cov) { check(g ? mc_group_set_coverage_hint(g, cov) : mc_set_coverage_hint(c, cov)); }
    void add_reads_file(const std::string &path, uint64_t *n) { check(g ? mc_group_add_reads_file(g, path.c_str(), n) : mc_add_reads_file(c, path.c_str(), n)); }
    void finalize(uint64_t *n) { check(g ? mc_group_finalize_counts(g, n) : mc_finalize_counts(c, n)); }
    void bfs_batch(const mc_bfs_job *jobs, uint32_t n, int cov, int64_t mk, int64_t mr, mc_bfs_result *out)
    {
        check(g ? mc_group_bfs_batch(g, jobs, n, cov, mk, mr, out) : mc_bfs_batch(c, jobs, n, cov, mk, mr, out));
    }
};

// the reads of all --reads files into the table; returns hm.size()
uint64_t load_reads(const Options &o, Engine &e)
{
    for (const std::string &path : o.reads) {
        const size_t slash = path.find_last_of('/');
        info("Loading file " + (slash == std::string::npos ? path : path.substr(slash + 1)) + "...");
        uint64_t n = 0;
        e.add_reads_file(path, &n);
        info(group_digits(n) + " reads added");
    }
    uint64_t n_distinct = 0;
    e.finalize(&n_distinct);
    info("Hashtable size: " + std::to_string(n_distinct) + " kmers");
    return n_distinct;
}

// --tool kmer-counter (src/tools/KmersCounter.java:56-121): count, then <name>.kmers.bin + <name>.stat.txt
int run_kmer_counter(const Options &o)
{
    if (o.k < 0) throw Error("Parameter 'k' is mandatory");
    if (o.reads.empty()) throw Error("Parameter 'reads' is mandatory");
    if (!open_work_dir(o, "k=" + std::to_string(o.k) + "\n")) return 0;
    const std::string out_dir = o.output_dir.empty() ? o.work_dir + "/kmers" : o.output_dir;
    int mode = MC_KEY_PACKED;
    if (o.k > 31) {  // (no --forcehash here: KmersCounter.java:59-69)
        info("Reading hashes of k-mers instead");
        std::string h = o.hash;
        for (char &c : h) c = (char)tolower((unsigned char)c);
        if (h == "fnv1a") { info("Using FNV1a hash function"); mode = MC_KEY_FNV1A; }
        else { info("Using default polynomial hash function"); mode = MC_KEY_POLY; }
    }
    mc_config cfg{};
    cfg.k = o.k;
    cfg.key_mode = mode;
    cfg.device = o.device;
    cfg.capacity_hint = o.capacity_hint;
    if (!o.devices.empty()) throw Error("--devices is for --tool environment-finder: kmer-counter writes one device's table (--device)");
    Engine E;
    E.open(cfg, {});
    mc_ctx *ctx = E.c;
    const uint64_t size = load_reads(o, E);
    // ReadersUtils.readDnaLazy(file).name(): the first file's name without its format extension
    std::string name = o.reads[0];
    const size_t slash = name.find_last_of('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    {
        std::string low = name;
        for (char &c : low) c = (char)tolower((unsigned char)c);
        for (const char *ext : {".fasta.gz", ".fa.gz", ".fn.gz", ".fna.gz", ".fastq.gz", ".fq.gz", ".fasta.bz2", ".fa.bz2", ".fn.bz2", ".fna.bz2",
                                ".fastq.bz2", ".fq.bz2", ".fasta", ".fa", ".fn", ".fna", ".fastq", ".fq", ".binq"}) {
            const size_t n = strlen(ext);
            if (low.size() >= n && low.compare(low.size() - n, n, ext) == 0) { name.resize(name.size() - n); break; }
        }
    }
    const std::string bin = out_dir + "/" + name + ".kmers.bin", st = out_dir + "/" + name + ".stat.txt";
    write_file(out_dir + "/.keep", "");
    remove((out_dir + "/.keep").c_str());
    logline("DEBUG", "Starting to print k-mers to " + bin);
    uint64_t total = 0, good = 0;
    MC_CHECK(ctx, mc_save_kmers(ctx, bin.c_str(), st.c_str(), 0, &total, &good));
    char pct[32];
    if (size) snprintf(pct, sizeof pct, "%.1f", good * 100.0 / (double)size);
    else snprintf(pct, sizeof pct, "NaN");  // (Java's String.format of 0.0 / 0)
    info(group_digits(size) + " k-mers found, " + group_digits(good) + " (" + pct + "%) of them is good (not erroneous)");
    if (size == 0) logline("WARN", "No k-mers found in reads! Perhaps you reads file is empty or k-mer size is too big");
    else if (good == 0 || good < (uint64_t)((double)size * 0.03))
        logline("WARN", "Too few good k-mers were found! Perhaps you should decrease k-mer size or --maximal-bad-frequency value");
    if (o.k <= 31) {
        const uint64_t all = (1ull << (2 * o.k)) / 2;  // (4^k)/2
        if (size == all) logline("WARN", "All possible k-mers were found in reads! Perhaps you should increase k-mer size");
        else if (size >= (uint64_t)((double)all * 0.99))
            logline("WARN", "Almost all possible k-mers were found in reads! Perhaps you should increase k-mer size");
    }
    info("k-mers printed to " + bin);
    write_file(o.work_dir + "/SUCCESS", "");
    return 0;
}

// --tool environment-finder-multi (src/tools/EnvironmentFinderMultiMain.java): no GPU involved
int run_multi(const Options &o)
{
    if (o.env.empty()) throw Error("Parameter 'env' is mandatory");
    if (o.seq.empty()) throw Error("Parameter 'seq' is mandatory");
    if (o.output.empty()) throw Error("Parameter 'output' is mandatory");
    if (!open_work_dir(o, "seq=" + o.seq + "\noutput=" + o.output + "\n")) return 0;
    const MultiResult r = environment_finder_multi(o.env, o.seq, o.geneid);
    write_multi(r, o.output);
    for (const std::string &l : r.log) logline(l.substr(0, 4).c_str(), l.substr(5));
    write_file(o.work_dir + "/SUCCESS", "");
    return 0;
}

int run(const Options &o)
{
    if (o.tool == "kmer-counter") return run_kmer_counter(o);
    if (o.tool == "environment-finder-multi") return run_multi(o);
    if (o.tool != "environment-finder")
        throw Error("Tool '" + o.tool + "' is not part of this build: only environment-finder, kmer-counter and environment-finder-multi are");
    if (o.k < 0) throw Error("Parameter 'k' is mandatory");
    if (o.seq.empty()) throw Error("Parameter 'seq' is mandatory");
    if (o.output.empty()) throw Error("Parameter 'output' is mandatory");
    if (o.maxkmers < 0 && o.maxradius < 0)  // EnvironmentFinderMain.java:171-175
        throw Error("At least one of --maxkmers and --maxradius parameters should be set");
    if (o.coverage < 0) throw Error("--coverage must not be negative (absent k-mers read as -1 and would pass)");
    if (o.k > 63)  // (the reference hashes k-mer STRINGS of any length for k > 31; the walk here keeps oriented k-mers in 128 bits)
        throw Error("k = " + std::to_string(o.k) + " is not supported: this build handles k <= 31 (packed keys) and 32 <= k <= 63 (hash keys)");

    const std::string wd = o.work_dir;
    if (!open_work_dir(o, "k=" + std::to_string(o.k) + "\nseq=" + o.seq + "\noutput=" + o.output + "\ncoverage=" +
                              std::to_string(o.coverage) + "\nbothdirs=" + (o.bothdirs ? "true" : "false") + "\n"))
        return 0;

    // loadInput (EnvironmentFinderMain.java:127-154)
    const bool hashed = o.k > 31 || o.forcehash;
    int mode = MC_KEY_PACKED;
    if (hashed) {
        info("Reading hashes of k-mers instead");
        std::string h = o.hash;
        for (char &c : h) c = (char)tolower((unsigned char)c);
        if (h == "fnv1a") { info("Using FNV1a hash function"); mode = MC_KEY_FNV1A; }
        else { info("Using default polynomial hash function"); mode = MC_KEY_POLY; }
    }
    mc_config cfg{};
    cfg.k = o.k;
    cfg.key_mode = mode;
    cfg.device = o.device;
    cfg.capacity_hint = o.capacity_hint;
    Engine E;
    E.open(cfg, o.devices);
    if (!o.devices.empty()) info("Counting on " + std::to_string(o.devices.size()) + " devices");
    E.set_coverage_hint(o.coverage);

    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t n_distinct = load_reads(o, E);
    logline("DEBUG", "k-mers HM size = " + group_digits(n_distinct));
    const auto t1 = std::chrono::steady_clock::now();

    SeedFile seeds;
    try {
        seeds = read_seed_fasta(o.seq);
    } catch (const Error &) {
        throw Error("Could not load sequences from " + o.seq);
    }
    SeedFile hic;
    std::vector<std::string> comments = seeds.comments;
    if (!o.hicseq.empty()) {
        try {
            hic = read_seed_fasta(o.hicseq);
        } catch (const Error &) {
            throw Error("Could not load Hi-C sequences from " + o.hicseq);
        }
        comments = hic.comments;  // the reference overwrites the comments (EnvironmentFinderMain.java:149)
    }

    // runImpl (:185-243): one calculator per sequence, or one for all with --merge
    struct Calc { std::string out_prefix; std::vector<std::string> bfs_seqs, genes; };
    std::vector<Calc> calcs;
    if (!o.merge) {
        for (size_t i = 0; i < seeds.dnas.size(); i++) {
            if (i >= comments.size()) throw Error("sequence " + std::to_string(i) + " has no FASTA comment to name its output directory");
            calcs.push_back(Calc{o.output + "/" + comments[i] + "/", {seeds.dnas[i]}, {seeds.dnas[i]}});
        }
    } else {
        info("hicSequences = " + std::to_string(hic.dnas.size()));
        Calc c{o.output + "/merged/", seeds.dnas, seeds.dnas};
        c.bfs_seqs.insert(c.bfs_seqs.end(), hic.dnas.begin(), hic.dnas.end());
        calcs.push_back(c);
    }
    const std::vector<int> dirs = pass_dirs(o.bothdirs);
    // The passes of up to CHUNK_JOBS / dirs calculators go to the GPU in one batch (the reference runs one
    // OneSequenceCalculator per sequence on a thread pool, EnvironmentFinderMain.java:218-225, without a limit on their
    // number): a chunk's environments are written and its results freed before the next one starts, so neither the job
    // count of mc_bfs_batch nor the memory of the per-job arrays grows with the number of seed sequences.
    constexpr size_t CHUNK_JOBS = 256;
    const size_t calcs_per_chunk = std::max<size_t>(1, CHUNK_JOBS / dirs.size());
    double bfs_ms = 0, out_ms = 0;
    unsigned long long bfs_rounds = 0, bfs_levels = 0;  // (metrics.json: how well the walks' look-ahead did -- levels per verification round)
    auto ms_between = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (size_t c0 = 0; c0 < calcs.size(); c0 += calcs_per_chunk) {
        const size_t c1 = std::min(calcs.size(), c0 + calcs_per_chunk);
        const auto tb0 = std::chrono::steady_clock::now();
        std::vector<std::vector<uint64_t>> shi(c1 - c0), slo(c1 - c0);
        std::vector<mc_bfs_job> jobs;
        for (size_t c = c0; c < c1; c++) {
            if (!o.merge) info("Finding environment for sequence " + shorten_label(calcs[c].bfs_seqs[0], o.k));
            else info("Finding single environment for " + std::to_string(seeds.dnas.size()) + " sequences");
            for (const std::string &s : calcs[c].bfs_seqs)
                for (size_t i = 0; i + (size_t)o.k <= s.size(); i++) {
                    uint64_t hi, lo;
                    pack_kmer(s.substr(i, (size_t)o.k), &hi, &lo);
                    shi[c - c0].push_back(hi);
                    slo[c - c0].push_back(lo);
                }
            for (int d : dirs) jobs.push_back(mc_bfs_job{shi[c - c0].data(), slo[c - c0].data(), shi[c - c0].size(), d});
        }
        std::vector<mc_bfs_result> res(jobs.size());
        struct ResGuard {  // (an exception below must not leak the library's result arrays)
            std::vector<mc_bfs_result> &r;
            ~ResGuard() { for (auto &x : r) mc_bfs_result_free(&x); }
        } guard{res};
        if (!jobs.empty()) E.bfs_batch(jobs.data(), (uint32_t)jobs.size(), o.coverage, o.maxkmers, o.maxradius, res.data());
        const auto tb1 = std::chrono::steady_clock::now();
        bfs_ms += ms_between(tb0, tb1);
        for (const mc_bfs_result &r : res) { bfs_rounds += r.rounds; bfs_levels += r.levels; }

        size_t j = 0;
        for (size_t c = c0; c < c1; c++) {
            Environment env(o.k, calcs[c].genes);
            bool fail = false;
            for (size_t d = 0; d < dirs.size(); d++, j++) {
                mc_bfs_result &r = res[j];
                if (r.n == 0) { fail = true; continue; }  // runBfs: queue.size() == 0 -> fail (:193-196)
                if (fail) continue;
                BfsPass p;
                p.dir = dirs[d];
                p.kmers.reserve(r.n);
                for (uint64_t i = 0; i < r.n; i++) p.kmers.push_back(((kmer_t)r.hi[i] << 64) | r.lo[i]);
                p.dist.assign(r.dist, r.dist + r.n);
                p.cov.assign(r.cov, r.cov + r.n);
                p.last.assign(r.last, r.last + r.n);
                env.add_pass(p, o.trim);
            }
            if (fail) {
                info("Could not find any k-mers of the target gene in the input, halting.");
                continue;
            }
            info("Extending endings by 0 kmers");  // extendEnvironment never adds anything
            if (!env.order_guaranteed())
                logline("WARN", "--trim removed k-mers from a treeified java.util.HashMap bin: line order of " + calcs[c].out_prefix +
                                " may differ from the JVM's inside that bin");
            env.write_all(calcs[c].out_prefix, o.chunklength);
        }
        out_ms += ms_between(tb1, std::chrono::steady_clock::now());
    }
    info("Finished processing all sequences!");

    mc_stats stt{};
    if (E.c) mc_get_stats(E.c, &stt);
    else if (E.g) mc_group_get_stats(E.g, &stt);  // (summed over the devices; times: the slowest device's)
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    char buf[512];
    snprintf(buf, sizeof buf,
             "{\"windows\": %llu, \"distinct_kmers\": %llu, \"load_and_count_ms\": %.3f, \"count_kernel_ms\": %.3f, "
             "\"bfs_ms\": %.3f, \"output_ms\": %.3f, \"table_bytes\": %llu, \"binned_runs\": %llu, \"bfs_levels\": %llu, \"bfs_rounds\": %llu}\n",
             (unsigned long long)stt.windows, (unsigned long long)n_distinct, ms(t0, t1), stt.count_total_ms, bfs_ms,
             out_ms, (unsigned long long)stt.table_bytes, (unsigned long long)stt.binned_runs, bfs_levels, bfs_rounds);  // (binned_runs: --devices, counting runs fed by the binned exchange)
    write_file(wd + "/metrics.json", buf);
    write_file(wd + "/SUCCESS", "");
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        if (argc <= 1) { usage(); return 0; }
        const Options o = parse_args(argc, argv);
        if (o.help) { usage(); return 0; }
        return run(o);
    } catch (const std::exception &e) {
        logline("ERROR", e.what());
        return 1;  // System.exit(1) on ExecutionFailedException, itmo!/utils/tool/Tool.java:450-462
    }
}
