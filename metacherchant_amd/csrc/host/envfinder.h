// Host side of the environment-finder path above the C ABI: seed reading, read ingest (the
// readers' N policy + 2-bit packing), runTrimPaths, the subgraph map, unitig compaction and the
// writers.  It mirrors the reference's classes (names below) so that the output files are
// byte-identical; the k-mer table and the BFS themselves live behind include/mcgpu.h.
//
// Citations: src/... = reference src/; itmo!/x = ru/ifmo/genetics/x in lib/itmo-assembler-src.jar.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace mch {

struct Error : std::runtime_error {  // plays ExecutionFailedException
    using std::runtime_error::runtime_error;
};

// ---- DNA strings (itmo!/dna/DnaTools.java:31,46-64,131-145; src/utils/StringUtils.java:8-41)
int code_of(char c);                       // A0 G1 C2 T3 (case-insensitive), -1 otherwise
std::string reverse_complement(const std::string &s);
std::string normalize_dna(const std::string &s);  // ASCII-lexicographic min of (s, rc(s))
std::vector<std::string> neighbors_by_dir(int dir, const std::string &kmer);  // A,G,C,T order; dir 0 interleaves L,R
void pack_kmer(const std::string &s, uint64_t *hi, uint64_t *lo);
std::string unpack_kmer(uint64_t hi, uint64_t lo, int k);

// ---- treeified bins of java.util.HashMap (JDK 8 HashMap.TreeNode).  A bin's iteration order is
// its nodes' next-chain, kept here as a vector of entry ids; the red-black tree over the same nodes decides where a new
// node is linked in: treeify() builds the tree in chain order and moves its root to the front (moveRootToFront),
// put() is putTreeVal (the new node goes right behind its tree parent, then balanceInsertion, then the root to the front).
// dir(x, p) = which way x goes below p: by the (signed) spread hash, then String.compareTo -- < 0 left, > 0 right.
class JavaTreeOrder {
public:
    explicit JavaTreeOrder(std::function<int(uint32_t, uint32_t)> dir) : dir_(std::move(dir)) {}
    void treeify(std::vector<uint32_t> &chain);
    void put(std::vector<uint32_t> &chain, uint32_t id);
    // removeTreeNode + balanceDeletion: the chain only loses `id` (with `movable` -- HashMap.remove; an iterator's remove, which
    // is what retainAll uses, passes false -- the root then moves to the front).  Returns true when the bin untreeifies (the tree
    // was too small BEFORE the removal) or is empty: the caller forgets the tree and keeps the chain as a plain list.
    bool remove(std::vector<uint32_t> &chain, uint32_t id, bool movable);
    void forget(const std::vector<uint32_t> &chain) { for (uint32_t id : chain) t_.erase(id); }

private:
    static constexpr uint32_t NIL = 0xFFFFFFFFu;
    struct Node { uint32_t parent = NIL, left = NIL, right = NIL; bool red = false; };
    uint32_t rotate_left(uint32_t root, uint32_t p);
    uint32_t rotate_right(uint32_t root, uint32_t p);
    uint32_t balance_insertion(uint32_t root, uint32_t x);
    uint32_t balance_deletion(uint32_t root, uint32_t x);
    static void root_to_front(std::vector<uint32_t> &chain, uint32_t root);
    std::function<int(uint32_t, uint32_t)> dir_;
    std::unordered_map<uint32_t, Node> t_;
};

// ---- java.util.HashMap<String,Integer> iteration order (JDK 8)
class JavaHashMap {
public:
    JavaHashMap();
    void put(const std::string &key, int value);  // new keys go to the tail of their bin; existing keep their place
    bool contains(const std::string &key) const { return index_.count(key) != 0; }
    int get(const std::string &key) const;         // throws if absent
    bool find(const std::string &key, int *value) const;
    void remove(const std::string &key, bool movable = false);
    size_t size() const { return size_; }
    // The order of a treeified bin is replayed node for node (JavaTreeOrder), removals included (removeTreeNode): nothing sets
    // the flag any more; it stays for callers that ask.
    bool treeified() const { return order_unknown_; }
    size_t bins_treeified() const { return n_treeified_; }  // bins that were treeified at some point (tests)
    template <typename F>
    void for_each(F &&f) const
    {
        for (const auto &bin : bins_)
            for (uint32_t e : bin) f(entries_[e].key, entries_[e].value);
    }

private:
    struct Entry { std::string key; int value; uint32_t hash; };
    void resize();
    int tree_dir(uint32_t x, uint32_t p) const;
    std::deque<Entry> entries_;
    std::vector<std::vector<uint32_t>> bins_;  // (a treeified bin's vector is its next-chain)
    std::vector<char> is_tree_;
    std::unordered_map<std::string, uint32_t> index_;
    JavaTreeOrder tree_;
    size_t cap_ = 16, size_ = 0, n_treeified_ = 0;
    bool order_unknown_ = false;
};

// ---- src/io/RichFastaReader.java:38-77 (+ DnaQ(String,0).toString(): N/n/. -> 'A', itmo!/dna/DnaQ.java:21-30)
struct SeedFile {
    std::vector<std::string> dnas, comments;
};
SeedFile read_seed_fasta(const std::string &path);  // throws Error when the file cannot be opened

// ---- read ingest: itmo!/io/ReadersUtils.java:27-53,104-121 format by extension;
// FASTA: records with N/n dropped whole (itmo!/io/readers/FastaReader.java:54-76);
// FASTQ: split at phred < 1, offset sniffed on the first 1000 records
// (itmo!/io/readers/FastqReader.java:53-112, FastaReaderFromXQSourceTrunc.java:61-95, ReadersUtils.java:57-77).
struct PackedBatch {
    std::vector<uint64_t> words;    // 2-bit packed, layout of include/mcgpu.h, with the pad word
    std::vector<uint64_t> offsets;  // n_reads + 1
    uint64_t n_reads() const { return offsets.empty() ? 0 : offsets.size() - 1; }
    void clear();
    void add_read(const char *s, size_t n);  // pure ACGT (any case); throws Error otherwise
    void finish();                           // appends the pad word
};
// Calls sink(batch) for every max_reads reads; returns the number of reads (pieces) delivered.
uint64_t load_reads_file(const std::string &path, size_t max_reads, const std::function<void(PackedBatch &)> &sink);

// What the device tokeniser (csrc/tokenizer.h) needs from the host: an uncompressed FASTA / FASTQ file mapped, its format,
// the quality offset of its first 1000 records, record starts to cut it at, and this file's own parser for a byte range
// the device declined (a byte that is no base, a record out of shape: the serial parser defines what happens then).
struct PlainReadsFile {
    const char *p = nullptr;
    size_t n = 0;
    bool fastq = false;
    int offset = 0;  // FASTQ: 33 or 64
    int fd = -1;
    PlainReadsFile() = default;
    PlainReadsFile(const PlainReadsFile &) = delete;
    PlainReadsFile &operator=(const PlainReadsFile &) = delete;
    ~PlainReadsFile();
};
// false: compressed, binq, unknown suffix, empty or unmappable, or a FASTQ whose first 1000 records are not plain
// four-part records -- load_reads_file() is the reader for those
bool map_plain_reads(const std::string &path, PlainReadsFile *out);
// first record start at or after q (the end of the file when there is none)
const char *plain_record_start(const PlainReadsFile &f, const char *q);
// the serial parser over [b, e), which must begin at a record start; returns the reads delivered
uint64_t parse_plain_range(const PlainReadsFile &f, const char *b, const char *e, size_t max_reads, const std::function<void(PackedBatch &)> &sink);

// ---- 2-bit packed k-mers (k <= 63): base i of the string is bits 2(k-1-i)+1..2(k-1-i), codes A0 G1 C2 T3,
// the layout of mc_bfs_result's hi/lo words (include/mcgpu.h)
typedef unsigned __int128 kmer_t;
kmer_t pack_kmer128(const std::string &s);
inline std::string unpack_kmer128(kmer_t v, int k) { return unpack_kmer((uint64_t)(v >> 64), (uint64_t)v, k); }
kmer_t reverse_complement128(kmer_t v, int k);
kmer_t normalize128(kmer_t v, int k);  // the packed form of normalize_dna(string)

// java.util.HashMap<String, Integer> over k-mer strings held packed: the same bins, resize points, in-bin
// order and treeify detection as JavaHashMap, with String.hashCode() evaluated on the characters the
// packed key stands for.  Entries are chained per bin the way the JDK chains them.
class JavaKmerMap {
public:
    explicit JavaKmerMap(int k);
    int put(kmer_t key, int value);   // entry index; an existing key keeps its place and gets the value
    int find_entry(kmer_t key) const;  // -1 when absent
    int value_at(int e) const { return entries_[(size_t)e].value; }
    void remove(kmer_t key, bool movable = false);
    size_t size() const { return size_; }
    size_t n_entries() const { return entries_.size(); }  // removed ones included: bound of the entry indices
    bool treeified() const { return order_unknown_; }  // (see JavaHashMap::treeified)
    size_t bins_treeified() const { return n_treeified_; }
    template <typename F>
    void for_each(F &&f) const  // f(key, value, entry index) in HashMap iteration order
    {
        for (uint32_t h : head_)
            for (uint32_t e = h; e != NIL; e = entries_[e].next) f(entries_[e].key, entries_[e].value, (int)e);
    }

private:
    static constexpr uint32_t NIL = 0xFFFFFFFFu;
    struct Entry { kmer_t key; int value; uint32_t hash, next; };
    uint32_t hash_of(kmer_t key) const;
    void resize();
    int tree_dir(uint32_t x, uint32_t p) const;
    void relink(size_t bin, const std::vector<uint32_t> &chain);  // head_/tail_/next of a bin from its chain
    int k_;
    std::vector<Entry> entries_;
    std::vector<uint32_t> head_, tail_;
    std::unordered_map<size_t, std::vector<uint32_t>> tree_bins_;  // treeified bins: their next-chains
    JavaTreeOrder tree_;
    size_t cap_ = 16, size_ = 0, n_treeified_ = 0;
    bool order_unknown_ = false;
};

// ---- one runBfs pass as delivered by mc_bfs_batch (or by a dump file in the CPU tests)
struct BfsPass {
    int dir = 0;
    std::vector<kmer_t> kmers;  // distanceToKmer insertion order
    std::vector<int32_t> dist;
    std::vector<int16_t> cov;
    std::vector<uint8_t> last;
};

// ---- src/algo/OneSequenceCalculator.java (after the BFS) + src/algo/SingleNode.java +
// src/io/writers/GFAWriter.java + src/io/writers/TSVWriter.java
class Environment {
public:
    Environment(int k, std::vector<std::string> gene_sequences);
    // :217-219 (+ runTrimPaths :241-262 when trim): distanceToKmer -> subgraph
    void add_pass(const BfsPass &p, bool trim);
    size_t size() const { return subgraph_.size(); }
    bool order_guaranteed() const { return !subgraph_.treeified() && !d_treeified_; }
    std::string graph_txt() const;                    // printEnvironment :297-310
    void create_picture();                            // initializeStructures + doMerge :387-451
    std::string seqs_fasta(int chunk_length) const;   // outputNodeSequences :354-385
    std::string graph_gfa() const;                    // GFAWriter.java:47-99
    std::string tsv_nodes() const;                    // TSVWriter.java:35-49
    std::string tsv_edges() const;                    // TSVWriter.java:51-79
    // writes graph.txt (+ the identical env.txt README.md:98 names), seqs.fasta, graph.gfa, tsvs/*
    void write_all(const std::string &out_prefix, int chunk_length);

private:
    struct Node {
        std::string sequence;
        int id;
        bool is_gene, deleted = false;
        int rc;                      // index of the reverse-complement node
        std::vector<int> neighbors;  // successors of rc(this), in node-array order
    };
    void merge_nodes(int first_plus, int second_minus);
    std::string node_id(const Node &n) const;
    int k_;
    std::vector<std::string> genes_;
    std::vector<kmer_t> gene_kmers_;  // sorted: every k-window of the gene sequences, for isGeneNode
    JavaKmerMap subgraph_;
    bool d_treeified_ = false;
    std::vector<Node> nodes_;
};

void write_file(const std::string &path, const std::string &text);  // mkdirs + write

// ---- --tool environment-finder-multi: src/tools/EnvironmentFinderMultiMain.java,
// src/algo/MultiSequenceCalculator.java, src/algo/MultiNode.java, src/io/writers/GFAWriterMulti.java,
// src/io/graph/DeBruijnGraphUtils.java.  CPU only: joins the graph.txt / env.txt files of several
// environment-finder runs into one coloured graph and two distance tables.
struct MultiResult {
    std::string seqs_fasta, graph_gfa, gene_fasta, jacard_sym, jacard_alt;  // <output>/seqs.fasta, graph.gfa, gene.fasta, Jacard_*.txt
    std::vector<std::string> log;                                            // "INFO ..." / "WARN ..." lines, in order
};
MultiResult environment_finder_multi(const std::vector<std::string> &env_paths, const std::string &seq_path, int gene_id);
void write_multi(const MultiResult &r, const std::string &output_dir);
std::string java_format_6_2f(float x);  // String.format("%6.2f", x)

}  // namespace mch
