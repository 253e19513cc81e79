// See envfinder.h.  Every function restates the reference lines cited there.
#include "envfinder.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <zlib.h>

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <cmath>
#include <functional>
#include <limits>
#include <set>
#include <sstream>
#include <thread>

namespace mch {

// ------------------------------------------------------------------------------------------ DNA strings

int code_of(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'G': case 'g': return 1;
    case 'C': case 'c': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

static inline char complement_char(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: throw Error(std::string("Incorrect nucleotide char: \"") + c + "\"");
    }
}

std::string reverse_complement(const std::string &s)
{
    std::string r(s.size(), 'A');
    for (size_t i = 0; i < s.size(); i++) r[i] = complement_char(s[s.size() - 1 - i]);
    return r;
}

std::string normalize_dna(const std::string &s)
{
    std::string rc = reverse_complement(s);
    return s.compare(rc) < 0 ? s : rc;  // String.compareTo on ASCII
}

std::vector<std::string> neighbors_by_dir(int dir, const std::string &kmer)
{
    static const char NUC[4] = {'A', 'G', 'C', 'T'};  // DnaTools.NUCLEOTIDES
    std::vector<std::string> out;
    const std::string head = kmer.substr(0, kmer.size() - 1), tail = kmer.substr(1);
    if (dir == -1) {
        for (char c : NUC) out.push_back(std::string(1, c) + head);
    } else if (dir == 1) {
        for (char c : NUC) out.push_back(tail + c);
    } else {
        for (char c : NUC) {
            out.push_back(std::string(1, c) + head);
            out.push_back(tail + c);
        }
    }
    return out;
}

void pack_kmer(const std::string &s, uint64_t *hi, uint64_t *lo)
{
    unsigned __int128 v = 0;
    for (char c : s) {
        const int code = code_of(c);
        if (code < 0) throw Error(std::string("Incorrect nucleotide char: \"") + c + "\"");
        v = (v << 2) | (unsigned)code;
    }
    *hi = (uint64_t)(v >> 64);
    *lo = (uint64_t)v;
}

std::string unpack_kmer(uint64_t hi, uint64_t lo, int k)
{
    const unsigned __int128 v = ((unsigned __int128)hi << 64) | lo;
    std::string s((size_t)k, 'A');
    for (int i = 0; i < k; i++) s[(size_t)i] = "AGCT"[(unsigned)(v >> (2 * (k - 1 - i))) & 3u];
    return s;
}

// ------------------------------------------------------------------------------------------ JavaTreeOrder
// java.util.HashMap.TreeNode (JDK 8): treeify, putTreeVal, balanceInsertion, rotateLeft / rotateRight, moveRootToFront.

uint32_t JavaTreeOrder::rotate_left(uint32_t root, uint32_t p)
{
    if (p == NIL) return root;
    const uint32_t r = t_[p].right;
    if (r == NIL) return root;
    const uint32_t rl = t_[p].right = t_[r].left;
    if (rl != NIL) t_[rl].parent = p;
    const uint32_t pp = t_[r].parent = t_[p].parent;
    if (pp == NIL) { root = r; t_[r].red = false; }
    else if (t_[pp].left == p) t_[pp].left = r;
    else t_[pp].right = r;
    t_[r].left = p;
    t_[p].parent = r;
    return root;
}

uint32_t JavaTreeOrder::rotate_right(uint32_t root, uint32_t p)
{
    if (p == NIL) return root;
    const uint32_t l = t_[p].left;
    if (l == NIL) return root;
    const uint32_t lr = t_[p].left = t_[l].right;
    if (lr != NIL) t_[lr].parent = p;
    const uint32_t pp = t_[l].parent = t_[p].parent;
    if (pp == NIL) { root = l; t_[l].red = false; }
    else if (t_[pp].right == p) t_[pp].right = l;
    else t_[pp].left = l;
    t_[l].right = p;
    t_[p].parent = l;
    return root;
}

uint32_t JavaTreeOrder::balance_insertion(uint32_t root, uint32_t x)
{
    t_[x].red = true;
    for (;;) {
        uint32_t xp = t_[x].parent;
        if (xp == NIL) { t_[x].red = false; return x; }
        uint32_t xpp = t_[xp].parent;
        if (!t_[xp].red || xpp == NIL) return root;
        const uint32_t xppl = t_[xpp].left;
        if (xp == xppl) {
            const uint32_t xppr = t_[xpp].right;
            if (xppr != NIL && t_[xppr].red) {
                t_[xppr].red = false; t_[xp].red = false; t_[xpp].red = true; x = xpp;
            } else {
                if (x == t_[xp].right) {
                    x = xp;
                    root = rotate_left(root, x);
                    xp = t_[x].parent;
                    xpp = xp == NIL ? NIL : t_[xp].parent;
                }
                if (xp != NIL) {
                    t_[xp].red = false;
                    if (xpp != NIL) { t_[xpp].red = true; root = rotate_right(root, xpp); }
                }
            }
        } else {
            if (xppl != NIL && t_[xppl].red) {
                t_[xppl].red = false; t_[xp].red = false; t_[xpp].red = true; x = xpp;
            } else {
                if (x == t_[xp].left) {
                    x = xp;
                    root = rotate_right(root, x);
                    xp = t_[x].parent;
                    xpp = xp == NIL ? NIL : t_[xp].parent;
                }
                if (xp != NIL) {
                    t_[xp].red = false;
                    if (xpp != NIL) { t_[xpp].red = true; root = rotate_left(root, xpp); }
                }
            }
        }
    }
}

void JavaTreeOrder::root_to_front(std::vector<uint32_t> &chain, uint32_t root)
{   // moveRootToFront: the root leaves its place in the chain and becomes its head
    if (chain.empty() || chain[0] == root) return;
    chain.erase(std::find(chain.begin(), chain.end(), root));
    chain.insert(chain.begin(), root);
}

void JavaTreeOrder::treeify(std::vector<uint32_t> &chain)
{
    uint32_t root = NIL;
    for (uint32_t x : chain) {
        t_[x] = Node{};
        if (root == NIL) { root = x; continue; }
        for (uint32_t p = root;;) {
            const int d = dir_(x, p);
            const uint32_t xp = p;
            p = d <= 0 ? t_[p].left : t_[p].right;
            if (p == NIL) {
                t_[x].parent = xp;
                if (d <= 0) t_[xp].left = x; else t_[xp].right = x;
                root = balance_insertion(root, x);
                break;
            }
        }
    }
    root_to_front(chain, root);
}

void JavaTreeOrder::put(std::vector<uint32_t> &chain, uint32_t id)
{   // putTreeVal for a key known to be absent
    uint32_t root = chain[0];
    while (t_[root].parent != NIL) root = t_[root].parent;
    for (uint32_t p = root;;) {
        const int d = dir_(id, p);
        const uint32_t xp = p;
        p = d <= 0 ? t_[p].left : t_[p].right;
        if (p == NIL) {
            t_[id] = Node{};
            t_[id].parent = xp;
            if (d <= 0) t_[xp].left = id; else t_[xp].right = id;
            chain.insert(std::find(chain.begin(), chain.end(), xp) + 1, id);  // xp.next = x
            root_to_front(chain, balance_insertion(root, id));
            return;
        }
    }
}

uint32_t JavaTreeOrder::balance_deletion(uint32_t root, uint32_t x)
{   // HashMap.TreeNode.balanceDeletion (JDK 8)
    auto red = [&](uint32_t n) { return n != NIL && t_[n].red; };
    for (;;) {
        if (x == NIL || x == root) return root;
        uint32_t xp = t_[x].parent;
        if (xp == NIL) { t_[x].red = false; return x; }
        if (t_[x].red) { t_[x].red = false; return root; }
        uint32_t xpl = t_[xp].left;
        if (xpl == x) {
            uint32_t xpr = t_[xp].right;
            if (red(xpr)) {
                t_[xpr].red = false; t_[xp].red = true;
                root = rotate_left(root, xp);
                xp = t_[x].parent;
                xpr = xp == NIL ? NIL : t_[xp].right;
            }
            if (xpr == NIL) { x = xp; continue; }
            uint32_t sl = t_[xpr].left, sr = t_[xpr].right;
            if (!red(sr) && !red(sl)) { t_[xpr].red = true; x = xp; continue; }
            if (!red(sr)) {
                if (sl != NIL) t_[sl].red = false;
                t_[xpr].red = true;
                root = rotate_right(root, xpr);
                xp = t_[x].parent;
                xpr = xp == NIL ? NIL : t_[xp].right;
            }
            if (xpr != NIL) {
                t_[xpr].red = xp == NIL ? false : t_[xp].red;
                sr = t_[xpr].right;
                if (sr != NIL) t_[sr].red = false;
            }
            if (xp != NIL) { t_[xp].red = false; root = rotate_left(root, xp); }
            x = root;
        } else {  // symmetric
            if (red(xpl)) {
                t_[xpl].red = false; t_[xp].red = true;
                root = rotate_right(root, xp);
                xp = t_[x].parent;
                xpl = xp == NIL ? NIL : t_[xp].left;
            }
            if (xpl == NIL) { x = xp; continue; }
            uint32_t sl = t_[xpl].left, sr = t_[xpl].right;
            if (!red(sl) && !red(sr)) { t_[xpl].red = true; x = xp; continue; }
            if (!red(sl)) {
                if (sr != NIL) t_[sr].red = false;
                t_[xpl].red = true;
                root = rotate_left(root, xpl);
                xp = t_[x].parent;
                xpl = xp == NIL ? NIL : t_[xp].left;
            }
            if (xpl != NIL) {
                t_[xpl].red = xp == NIL ? false : t_[xp].red;
                sl = t_[xpl].left;
                if (sl != NIL) t_[sl].red = false;
            }
            if (xp != NIL) { t_[xp].red = false; root = rotate_right(root, xp); }
            x = root;
        }
    }
}

bool JavaTreeOrder::remove(std::vector<uint32_t> &chain, uint32_t p, bool movable)
{   // HashMap.TreeNode.removeTreeNode (JDK 8)
    uint32_t root = chain[0];  // ("first" as it was before the node left the chain)
    chain.erase(std::find(chain.begin(), chain.end(), p));
    if (chain.empty()) { t_.erase(p); return true; }
    while (t_[root].parent != NIL) root = t_[root].parent;
    {
        const uint32_t rl = t_[root].left;
        if (t_[root].right == NIL || rl == NIL || t_[rl].left == NIL) { t_.erase(p); return true; }  // too small: untreeify
    }
    const uint32_t pl = t_[p].left, pr = t_[p].right;
    uint32_t replacement;
    if (pl != NIL && pr != NIL) {
        uint32_t s = pr;
        while (t_[s].left != NIL) s = t_[s].left;  // the successor
        std::swap(t_[s].red, t_[p].red);
        const uint32_t sr = t_[s].right, pp = t_[p].parent;
        if (s == pr) {  // p was s's direct parent
            t_[p].parent = s;
            t_[s].right = p;
        } else {
            const uint32_t sp = t_[s].parent;
            t_[p].parent = sp;
            if (sp != NIL) { if (s == t_[sp].left) t_[sp].left = p; else t_[sp].right = p; }
            t_[s].right = pr;
            if (pr != NIL) t_[pr].parent = s;
        }
        t_[p].left = NIL;
        t_[p].right = sr;
        if (sr != NIL) t_[sr].parent = p;
        t_[s].left = pl;
        if (pl != NIL) t_[pl].parent = s;
        t_[s].parent = pp;
        if (pp == NIL) root = s;
        else if (p == t_[pp].left) t_[pp].left = s;
        else t_[pp].right = s;
        replacement = sr != NIL ? sr : p;
    } else if (pl != NIL) replacement = pl;
    else if (pr != NIL) replacement = pr;
    else replacement = p;
    if (replacement != p) {
        const uint32_t pp = t_[replacement].parent = t_[p].parent;
        if (pp == NIL) root = replacement;
        else if (p == t_[pp].left) t_[pp].left = replacement;
        else t_[pp].right = replacement;
        t_[p].left = t_[p].right = t_[p].parent = NIL;
    }
    const uint32_t r = t_[p].red ? root : balance_deletion(root, replacement);
    if (replacement == p) {  // detach
        const uint32_t pp = t_[p].parent;
        t_[p].parent = NIL;
        if (pp != NIL) {
            if (p == t_[pp].left) t_[pp].left = NIL;
            else if (p == t_[pp].right) t_[pp].right = NIL;
        }
    }
    t_.erase(p);
    if (movable) root_to_front(chain, r);
    return false;
}

// ------------------------------------------------------------------------------------------ JavaHashMap

static uint32_t java_string_hash(const std::string &s)
{
    uint32_t h = 0;
    for (unsigned char c : s) h = 31u * h + c;
    return h;
}

JavaHashMap::JavaHashMap() : bins_(16), is_tree_(16, 0), tree_([this](uint32_t x, uint32_t p) { return tree_dir(x, p); }) {}

int JavaHashMap::tree_dir(uint32_t x, uint32_t p) const
{   // TreeNode order: the (signed) spread hash, then String.compareTo (distinct keys never compare equal)
    const int32_t h = (int32_t)entries_[x].hash, ph = (int32_t)entries_[p].hash;
    if (ph > h) return -1;
    if (ph < h) return 1;
    return entries_[x].key < entries_[p].key ? -1 : 1;
}

void JavaHashMap::put(const std::string &key, int value)
{
    auto it = index_.find(key);
    if (it != index_.end()) {
        entries_[it->second].value = value;
        return;
    }
    uint32_t h = java_string_hash(key);
    h ^= h >> 16;
    const uint32_t e = (uint32_t)entries_.size();
    entries_.push_back(Entry{key, value, h});
    index_.emplace(key, e);
    const size_t b = h & (cap_ - 1);
    auto &bin = bins_[b];
    if (is_tree_[b]) {
        tree_.put(bin, e);
    } else {
        bin.push_back(e);
        if (bin.size() >= 9) {  // a 9th node: treeifyBin (or a resize while the table is smaller than 64)
            if (cap_ >= 64) { tree_.treeify(bin); is_tree_[b] = 1; n_treeified_++; } else resize();
        }
    }
    size_++;
    if ((double)size_ > 0.75 * (double)cap_) resize();
}

void JavaHashMap::resize()
{
    const size_t ocap = cap_, ncap = cap_ * 2;
    std::vector<std::vector<uint32_t>> nb(ncap);
    std::vector<char> nt(ncap, 0);
    for (size_t j = 0; j < ocap; j++) {
        const auto &bin = bins_[j];
        for (uint32_t e : bin) nb[entries_[e].hash & (ncap - 1)].push_back(e);  // lo/hi split keeps relative order
        if (!is_tree_[j]) continue;
        // TreeNode.split: halves of <= 6 nodes become plain lists again; the others stay trees, rebuilt (treeify: the
        // root moves to the front) only when the bin really split
        auto &lo = nb[j], &hi = nb[j + ocap];
        const bool both = !lo.empty() && !hi.empty();
        for (auto *half : {&lo, &hi}) {
            if (half->empty()) continue;
            const size_t at = half == &lo ? j : j + ocap;
            if (half->size() <= 6) tree_.forget(*half);
            else { nt[at] = 1; if (both) tree_.treeify(*half); }
        }
    }
    cap_ = ncap;
    bins_.swap(nb);
    is_tree_.swap(nt);
}

int JavaHashMap::get(const std::string &key) const
{
    auto it = index_.find(key);
    if (it == index_.end()) throw Error("JavaHashMap::get: missing key " + key);
    return entries_[it->second].value;
}

bool JavaHashMap::find(const std::string &key, int *value) const
{
    auto it = index_.find(key);
    if (it == index_.end()) return false;
    *value = entries_[it->second].value;
    return true;
}

void JavaHashMap::remove(const std::string &key, bool movable)
{   // (movable: HashMap.remove(key) passes true, an iterator's remove -- retainAll -- false)
    auto it = index_.find(key);
    if (it == index_.end()) return;
    const size_t b = entries_[it->second].hash & (cap_ - 1);
    auto &bin = bins_[b];
    if (is_tree_[b]) {  // TreeNode.removeTreeNode
        if (tree_.remove(bin, it->second, movable)) { tree_.forget(bin); is_tree_[b] = 0; }
    } else {
        bin.erase(std::find(bin.begin(), bin.end(), it->second));
    }
    index_.erase(it);
    size_--;
}

// ------------------------------------------------------------------------------------------ files

static void mkdirs(const std::string &dir)
{
    if (dir.empty()) return;
    std::string cur;
    for (size_t i = 0; i <= dir.size(); i++) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && cur != "/") {
                if (mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
                    throw Error("Could not create directory " + cur + ": " + strerror(errno));
            }
        }
        if (i < dir.size()) cur.push_back(dir[i]);
    }
}

void write_file(const std::string &path, const std::string &text)
{
    const size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) mkdirs(path.substr(0, slash));
    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("Could not write " + path);
    f << text;
}

static bool read_lines(const std::string &path, std::vector<std::string> *lines)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        lines->push_back(line);
    }
    return true;
}

// ------------------------------------------------------------------------------------------ seeds

static std::string dnaq_string(const std::string &s)
{
    std::string r;
    r.reserve(s.size());
    for (char c : s) {
        if (c == 'N' || c == 'n' || c == '.') {
            r.push_back('A');  // DnaQ: unknown -> nuc 0
        } else {
            const int code = code_of(c);
            if (code < 0) throw Error(std::string("Incorrect nucleotide char: \"") + c + "\"");
            r.push_back("AGCT"[code]);
        }
    }
    return r;
}

SeedFile read_seed_fasta(const std::string &path)
{
    std::vector<std::string> lines;
    if (!read_lines(path, &lines)) throw Error("cannot open " + path);
    SeedFile out;
    bool last_comment = true;
    std::string cur_comment, cur_dna;
    for (const std::string &line : lines) {
        if (!line.empty() && (line[0] == '>' || line[0] == ';')) {
            if (!last_comment) {
                out.dnas.push_back(dnaq_string(cur_dna));
                cur_dna.clear();
                cur_comment.clear();
            }
            cur_comment += line.substr(1);
            last_comment = true;
        } else {
            if (last_comment) {
                out.comments.push_back(cur_comment);
                cur_dna.clear();
                cur_comment.clear();
            }
            cur_dna += line;
            last_comment = false;
        }
    }
    if (!cur_comment.empty()) out.comments.push_back(cur_comment);
    if (!cur_dna.empty()) out.dnas.push_back(dnaq_string(cur_dna));
    return out;
}

// ------------------------------------------------------------------------------------------ reads

void PackedBatch::clear()
{
    words.clear();
    offsets.assign(1, 0);
}

namespace {
struct CodeLut {  // A0 G1 C2 T3 (either case), 0xFF otherwise
    uint8_t v[256];
    CodeLut()
    {
        memset(v, 0xFF, sizeof v);
        v[(unsigned char)'A'] = v[(unsigned char)'a'] = 0;
        v[(unsigned char)'G'] = v[(unsigned char)'g'] = 1;
        v[(unsigned char)'C'] = v[(unsigned char)'c'] = 2;
        v[(unsigned char)'T'] = v[(unsigned char)'t'] = 3;
    }
};
const CodeLut CODE_LUT;
}  // namespace

void PackedBatch::add_read(const char *s, size_t n)
{
    if (offsets.empty()) offsets.assign(1, 0);
    uint64_t pos = offsets.back();
    words.resize((pos + n + 31) / 32, 0);
    // a word at a time: the open word is completed, then whole words of 32 bases, then the tail
    size_t i = 0;
    uint32_t bad = 0;
    uint64_t *w = words.data();
    while (i < n) {
        const uint32_t in_word = (uint32_t)(pos & 31), take = (uint32_t)std::min<size_t>(32 - in_word, n - i);
        uint64_t acc = 0;
        for (uint32_t j = 0; j < take; j++) {
            const uint8_t c = CODE_LUT.v[(unsigned char)s[i + j]];
            bad |= c;
            acc = (acc << 2) | (c & 3u);
        }
        w[pos >> 5] |= acc << (2 * (32 - in_word - take));
        pos += take;
        i += take;
    }
    if (bad & 0x80) {
        for (size_t q = 0; q < n; q++)
            if (code_of(s[q]) < 0)
                throw Error(std::string("read contains the character '") + s[q] +
                            "': IUPAC codes other than N are replaced at random by the reference "
                            "(itmo!/dna/DnaTools.java:66-117), which has no defined result; rejecting the input");
    }
    offsets.push_back(pos);
}

void PackedBatch::finish()
{
    if (offsets.empty()) offsets.assign(1, 0);
    words.resize((offsets.back() + 31) / 32 + 1, 0);
}

static bool ends_with(const std::string &s, const char *suf)
{
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

static std::string lower(std::string s)
{
    for (char &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

// Bytes of a file, plain, gzip or bzip2 (itmo!/io/readers/FastaGZReader.java, FastqGZReader.java: the same readers
// over a java.util.zip.GZIPInputStream; FastaBZ2Reader.java:27, FastqBZ2Reader.java: over Hadoop's BZip2Codec
// stream; concatenated members / streams read on, as there).  bzip2 goes through the system's libbz2.so.1.0, loaded
// when the first .bz2 file is opened: its 1.0 ABI (bz_stream + BZ2_bzDecompress*) is declared here because the
// image ships the library without its header.
namespace {
enum Compression { COMP_NONE, COMP_GZ, COMP_BZ2 };

struct Bz2Api {
    struct Stream {
        char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
        char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
        void *state;
        void *(*bzalloc)(void *, int, int);
        void (*bzfree)(void *, void *);
        void *opaque;
    };
    int (*init)(Stream *, int, int) = nullptr;
    int (*run)(Stream *) = nullptr;
    int (*end)(Stream *) = nullptr;
    static const Bz2Api &get()
    {
        static const Bz2Api api = [] {
            Bz2Api a;
            void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                a.init = (int (*)(Stream *, int, int))dlsym(h, "BZ2_bzDecompressInit");
                a.run = (int (*)(Stream *))dlsym(h, "BZ2_bzDecompress");
                a.end = (int (*)(Stream *))dlsym(h, "BZ2_bzDecompressEnd");
            }
            return a;
        }();
        if (!api.init || !api.run || !api.end) throw Error("bzip2 input needs libbz2.so.1.0, which could not be loaded");
        return api;
    }
};

class ByteSource {
public:
    ByteSource(const std::string &path, Compression comp) : comp_(comp)
    {
        if (comp_ == COMP_GZ) {
            g_ = gzopen(path.c_str(), "rb");
            if (!g_) throw Error("Failed to read from file " + path);
            gzbuffer(g_, 1u << 20);
        } else {
            f_ = fopen(path.c_str(), "rb");
            if (!f_) throw Error("Failed to read from file " + path);
            if (comp_ == COMP_BZ2) {
                bz_ = &Bz2Api::get();
                in_.resize(1u << 20);
                memset(&bs_, 0, sizeof bs_);
            }
        }
    }
    ~ByteSource()
    {
        if (g_) gzclose(g_);
        if (bz_open_) bz_->end(&bs_);
        if (f_) fclose(f_);
    }
    ByteSource(const ByteSource &) = delete;
    ByteSource &operator=(const ByteSource &) = delete;
    // up to n bytes; 0 at the end of the data
    size_t read(char *dst, size_t n)
    {
        if (comp_ == COMP_NONE) return fread(dst, 1, n, f_);
        if (comp_ == COMP_GZ) {
            const int got = gzread(g_, dst, (unsigned)std::min<size_t>(n, 1u << 30));
            if (got < 0) throw Error("Failed to decompress the input (corrupt gzip stream)");
            return (size_t)got;
        }
        size_t done = 0;
        while (done < n && !bz_eof_) {
            if (bs_.avail_in == 0) {
                bs_.next_in = in_.data();
                bs_.avail_in = (unsigned)fread(in_.data(), 1, in_.size(), f_);
                if (bs_.avail_in == 0) {
                    if (bz_open_) throw Error("Failed to decompress the input (truncated bzip2 stream)");
                    bz_eof_ = true;  // the file ends between two streams
                    break;
                }
            }
            if (!bz_open_) {
                if (bz_->init(&bs_, 0, 0) != 0) throw Error("Failed to decompress the input (bzip2 initialisation)");
                bz_open_ = true;
            }
            bs_.next_out = dst + done;
            bs_.avail_out = (unsigned)std::min<size_t>(n - done, 1u << 30);
            const unsigned before = bs_.avail_out;
            const int rc = bz_->run(&bs_);
            done += before - bs_.avail_out;
            if (rc == 4) {  // BZ_STREAM_END: another stream may follow
                bz_->end(&bs_);
                bz_open_ = false;
            } else if (rc != 0) {
                throw Error("Failed to decompress the input (corrupt bzip2 stream)");
            }
        }
        return done;
    }

private:
    Compression comp_;
    gzFile g_ = nullptr;
    FILE *f_ = nullptr;
    const Bz2Api *bz_ = nullptr;
    Bz2Api::Stream bs_;
    std::vector<char> in_;
    bool bz_open_ = false, bz_eof_ = false;
};

// Lines of such a file ('\n' ends a line, one trailing '\r' comes off)
class LineSource {
public:
    LineSource(const std::string &path, Compression comp) : src_(path, comp), buf_(1u << 20) {}
    bool getline(std::string &l)
    {
        l.clear();
        bool any = false;
        for (;;) {
            if (pos_ == len_) {
                len_ = src_.read(buf_.data(), buf_.size());
                pos_ = 0;
                if (len_ == 0) {
                    if (!any) return false;
                    break;
                }
            }
            any = true;
            const char *b = buf_.data() + pos_;
            const char *nl = (const char *)memchr(b, '\n', len_ - pos_);
            if (nl) {
                l.append(b, (size_t)(nl - b));
                pos_ += (size_t)(nl - b) + 1;
                break;
            }
            l.append(b, len_ - pos_);
            pos_ = len_;
        }
        if (!l.empty() && l.back() == '\r') l.pop_back();
        return true;
    }

private:
    ByteSource src_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
};
}  // namespace

namespace {

// lines of a memory range (a chunk of a memory-mapped file)
class MemLines {
public:
    MemLines(const char *b, const char *e) : p_(b), end_(e) {}
    bool getline(std::string &l)
    {
        if (p_ >= end_) return false;
        const char *nl = static_cast<const char *>(memchr(p_, '\n', (size_t)(end_ - p_)));
        const char *stop = nl ? nl : end_;
        l.assign(p_, (size_t)(stop - p_));
        p_ = nl ? nl + 1 : end_;
        if (!l.empty() && l.back() == '\r') l.pop_back();
        return true;
    }

private:
    const char *p_, *end_;
};

// itmo!/io/readers/FastaReader.java:54-104: multi-line records concatenated, records with N / n dropped whole
template <class Src, class Emit>
void parse_fasta(Src &src, Emit &&emit)
{
    std::string line, sb;
    auto flush = [&] {
        if (!sb.empty() && sb.find('N') == std::string::npos && sb.find('n') == std::string::npos) emit(sb.data(), sb.size());
        sb.clear();
    };
    while (src.getline(line)) {
        if (!line.empty() && (line[0] == '>' || line[0] == ';')) {
            if (!sb.empty()) flush();
        } else if (sb.empty()) {
            sb.swap(line);  // (the usual single-line record is not copied again)
        } else {
            sb += line;
        }
    }
    flush();
}

// itmo!/io/readers/FastqReader.java:53-112 + FastaReaderFromXQSourceTrunc.java:61-95 + itmo!/io/ReadersUtils.java:57-77.
// Records: marker line ("@id" / "+id") + content line, twice; empty lines before a marker are skipped.
// offset < 0: the quality offset is sniffed on the first 1000 records (Illumina+64 unless a char < 64 shows up).
// strict: the first marker must be '@' and the second '+' (what the parallel reader relies on); returns false otherwise.
template <class Src, class Emit>
bool parse_fastq(Src &src, Emit &&emit, int offset, bool strict, int *sniffed = nullptr, size_t sniff_only = 0)
{
    char marker = 0;
    auto next_data = [&](std::string &out) -> bool {
        std::string l;
        for (;;) {
            if (!src.getline(l)) return false;
            if (!l.empty()) break;
        }
        if (l[0] != '@' && l[0] != '+') throw Error("Unknown structure of fastq file! Waiting \"@ID\" or \"+ID\" string");
        marker = l[0];
        if (!src.getline(out)) throw Error("Unexpected end of file. File is corrupted/Format mismatch.");
        return true;
    };
    std::vector<std::pair<std::string, std::string>> head;  // the records the offset is sniffed on
    auto process = [&](const std::string &d, const std::string &q) {
        // truncateByQuality(1): a base with phred < 1 (or N n .) ends the piece and is dropped; pieces go out straight
        // from the line
        size_t start = 0;
        const size_t n = d.size();
        for (size_t i = 0; i < n; i++) {
            const unsigned char c = (unsigned char)d[i];
            bool bad = c == 'N' || c == 'n' || c == '.';
            if (!bad) {
                const int qc = (unsigned char)q[i];
                if (qc < offset || qc > 126) throw Error("Invalid quality code char");
                bad = ((qc - offset) & 63) < 1;  // the phred lives in 6 bits of a byte (DnaQBuilder.java:32-35): 64 wraps to 0
            }
            if (bad) {
                if (i > start) emit(d.data() + start, i - start);
                start = i + 1;
            }
        }
        if (n > start) emit(d.data() + start, n - start);
    };
    auto sniff = [&] {
        offset = 64;
        for (const auto &r : head)
            for (size_t i = 0; i < r.first.size(); i++) {
                if (r.first[i] == 'N' || r.first[i] == 'n' || r.first[i] == '.') continue;
                const int qc = (unsigned char)r.second[i];
                if (qc < 64 || qc > 126) { offset = 33; return; }
            }
    };
    std::string d, q;
    while (next_data(d)) {
        if (strict && marker != '@') return false;
        if (!next_data(q)) throw Error("Unexpected end of file. File is corrupted/Format mismatch.");
        if (strict && marker != '+') return false;
        if (d.size() != q.size()) throw Error("Bad DnaQ record: length of chars and quality is not the same.");
        if (offset < 0) {
            head.emplace_back(d, q);
            if (head.size() == 1000) {
                sniff();
                if (sniff_only) { *sniffed = offset; return true; }
                for (const auto &r : head) process(r.first, r.second);
                head.clear();
            }
        } else {
            process(d, q);
        }
    }
    if (offset < 0) {
        sniff();
        if (sniff_only) { *sniffed = offset; return true; }
        for (const auto &r : head) process(r.first, r.second);
    }
    return true;
}

// itmo!/io/readers/BinqReader.java:52-86: records of a 4-byte big-endian length and that many bytes (phred << 2 | nuc);
// 0xFF bytes in front of a record are padding.  Pieces as in parse_fastq (FastaReaderFromXQSourceTrunc.java:61-95).
template <class Src, class Emit>
void parse_binq(Src &src, Emit &&emit, const std::string &file_name)
{
    std::vector<char> buf(1u << 20);
    size_t pos = 0, len = 0;
    auto next_byte = [&]() -> int {
        if (pos == len) {
            len = src.read(buf.data(), buf.size());
            pos = 0;
            if (len == 0) return -1;
        }
        return (unsigned char)buf[pos++];
    };
    std::string rec, piece;
    for (;;) {
        int b0 = next_byte();
        while (b0 == 255) b0 = next_byte();
        if (b0 < 0) return;
        const int b1 = next_byte(), b2 = next_byte(), b3 = next_byte();
        if (b1 < 0 || b2 < 0 || b3 < 0) throw Error("Unexpected end of file " + file_name);
        const size_t n = ((size_t)b0 << 24) + ((size_t)b1 << 16) + ((size_t)b2 << 8) + (size_t)b3;
        rec.resize(n);
        for (size_t got = 0; got < n;) {
            if (pos == len) {
                len = src.read(buf.data(), buf.size());
                pos = 0;
                if (len == 0) throw Error("Unexpected end of file " + file_name);
            }
            const size_t take = std::min(n - got, len - pos);
            memcpy(&rec[got], buf.data() + pos, take);
            got += take;
            pos += take;
        }
        piece.clear();
        for (size_t i = 0; i < n; i++) {
            const unsigned v = (unsigned char)rec[i];
            if ((v >> 2) < 1) {  // truncateByQuality(1)
                if (!piece.empty()) emit(piece.data(), piece.size());
                piece.clear();
            } else {
                piece.push_back("AGCT"[v & 3]);
            }
        }
        if (!piece.empty()) emit(piece.data(), piece.size());
    }
}

struct Mapped {  // a read-only memory map of a whole file
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped()
    {
        if (p && n) munmap(const_cast<char *>(p), n);
        if (fd >= 0) close(fd);
    }
};

size_t env_size(const char *name, size_t dflt)
{
    const char *e = getenv(name);
    return e && *e ? (size_t)strtoull(e, nullptr, 10) : dflt;
}

// Parallel ingest of an uncompressed file: cut at record starts, parse the chunks on all cores, hand the batches over
// in file order.  Returns false (nothing delivered) when the file is small, cannot be mapped, or does not have the
// plain structure the cutting relies on -- the caller then reads it serially, which defines the behaviour.
bool load_reads_parallel(const std::string &path, bool fastq, size_t max_reads, const std::function<void(PackedBatch &)> &sink,
                         uint64_t *delivered)
{
    const bool dbg = getenv("MC_INGEST_DEBUG") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = tnow();
    size_t n_threads = env_size("MC_INGEST_THREADS", std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 128));
    const size_t min_chunk = std::max<size_t>(env_size("MC_INGEST_CHUNK_BYTES", 4u << 20), 64);
    if (n_threads < 2) return false;
    Mapped m;
    m.fd = open(path.c_str(), O_RDONLY);
    if (m.fd < 0) return false;
    struct stat st;
    if (fstat(m.fd, &st) != 0 || st.st_size < (off_t)(2 * min_chunk)) return false;
    m.n = (size_t)st.st_size;
    void *mp = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (mp == MAP_FAILED) { m.n = 0; return false; }
    m.p = static_cast<const char *>(mp);
    n_threads = std::min(n_threads, m.n / min_chunk);
    const char *begin = m.p, *end = m.p + m.n;
    auto line_start_after = [&](const char *q) -> const char * {  // first line start at or after q
        if (q <= begin) return begin;
        const char *nl = static_cast<const char *>(memchr(q - 1, '\n', (size_t)(end - (q - 1))));
        return nl ? nl + 1 : end;
    };
    auto line_end = [&](const char *q) { const char *nl = static_cast<const char *>(memchr(q, '\n', (size_t)(end - q))); return nl ? nl : end; };
    auto next_line = [&](const char *q) { const char *e = line_end(q); return e < end ? e + 1 : end; };
    auto len_of = [&](const char *q) { const char *e = line_end(q); size_t l = (size_t)(e - q); if (l && q[l - 1] == '\r') l--; return l; };
    // record start at or after q: FASTA a header line; FASTQ '@' line whose third line starts with '+' and whose second
    // and fourth lines have the same length (a quality line that starts with '@' fails the '+' test)
    auto record_start = [&](const char *q) -> const char * {
        const char *l = line_start_after(q);
        for (int guard = 0; l < end && guard < 100000; guard++, l = next_line(l)) {
            if (!fastq) {
                if (*l == '>' || *l == ';') return l;
                continue;
            }
            if (*l != '@') continue;
            const char *l2 = next_line(l), *l3 = l2 < end ? next_line(l2) : end;
            if (l3 >= end || *l3 != '+') continue;
            const char *l4 = next_line(l3);
            if (l4 < end && len_of(l2) == len_of(l4)) return l;
        }
        return end;
    };
    std::vector<const char *> cut(n_threads + 1);
    cut[0] = begin;
    cut[n_threads] = end;
    for (size_t t = 1; t < n_threads; t++) cut[t] = std::max(cut[t - 1], record_start(begin + m.n / n_threads * t));
    int offset = -1;
    if (fastq) {  // the quality offset comes from the first 1000 records of the file
        MemLines src(begin, end);
        if (!parse_fastq(src, [](const char *, size_t) {}, -1, true, &offset, 1) || offset < 0) return false;
    }
    const double t_cut = tnow();
    struct Part { std::vector<PackedBatch> batches; uint64_t reads = 0; bool ok = true; };
    std::vector<Part> parts(n_threads);
    std::vector<std::thread> threads;
    for (size_t t = 0; t < n_threads; t++)
        threads.emplace_back([&, t] {
            Part &P = parts[t];
            try {
                PackedBatch batch;
                batch.clear();
                // room for the whole chunk up front: growing vectors map and unmap memory, which serialises the threads
                const size_t chunk_bytes = (size_t)(cut[t + 1] - cut[t]);
                batch.words.reserve(chunk_bytes / 32 + 16);
                batch.offsets.reserve(std::min<size_t>(max_reads, chunk_bytes / 64) + 16);
                auto emit = [&](const char *s, size_t n) {
                    batch.add_read(s, n);
                    P.reads++;
                    if (batch.n_reads() >= max_reads) {
                        batch.finish();
                        P.batches.push_back(std::move(batch));
                        batch = PackedBatch();
                        batch.clear();
                    }
                };
                MemLines src(cut[t], cut[t + 1]);
                if (fastq) P.ok = parse_fastq(src, emit, offset, true);
                else parse_fasta(src, emit);
                if (batch.n_reads() > 0) {
                    batch.finish();
                    P.batches.push_back(std::move(batch));
                }
            } catch (...) {
                P.ok = false;  // the serial reader will meet the same problem and report it
            }
        });
    for (auto &th : threads) th.join();
    const double t_parse = tnow();
    for (const Part &P : parts)
        if (!P.ok) return false;
    *delivered = 0;
    for (const Part &P : parts) *delivered += P.reads;
    // Every thread leaves a small batch; handing them over one by one would cost a device launch each.  They are
    // joined, in file order, into batches of up to max_reads reads: the pieces' words are shifted into place in
    // parallel (a piece starts wherever the previous one ended, not at a word boundary).
    std::vector<PackedBatch *> pieces;
    for (Part &P : parts)
        for (PackedBatch &b : P.batches)
            if (b.n_reads()) pieces.push_back(&b);
    size_t first = 0;
    while (first < pieces.size()) {
        size_t last = first;
        uint64_t reads = 0, bases = 0;
        std::vector<uint64_t> base_at;  // where each piece starts, in bases
        while (last < pieces.size() && (last == first || reads + pieces[last]->n_reads() <= max_reads)) {
            base_at.push_back(bases);
            reads += pieces[last]->n_reads();
            bases += pieces[last]->offsets.back();
            last++;
        }
        if (last - first == 1) {
            sink(*pieces[first]);
            first = last;
            continue;
        }
        const double t_m0 = tnow();
        PackedBatch out;
        out.words.assign((bases + 31) / 32 + 1, 0);
        out.offsets.resize(reads + 1);
        out.offsets[0] = 0;
        std::vector<uint64_t> read_at(last - first);
        {
            uint64_t r = 0;
            for (size_t i = first; i < last; i++) { read_at[i - first] = r; r += pieces[i]->n_reads(); }
        }
        std::vector<std::thread> mergers;
        const size_t n_merge = std::min<size_t>(last - first, n_threads);
        for (size_t w = 0; w < n_merge; w++)
            mergers.emplace_back([&, w] {
                for (size_t i = first + w; i < last; i += n_merge) {
                    const PackedBatch &b = *pieces[i];
                    const uint64_t at = base_at[i - first], n_src = (b.offsets.back() + 31) / 32;
                    const uint64_t dw = at / 32, sh = 2 * (at % 32);
                    // only the first and the last destination words of a piece can be shared with a neighbour
                    auto put = [&](uint64_t idx, uint64_t v) {
                        if (!v) return;
                        if (idx == dw || idx + 1 >= dw + n_src) __atomic_fetch_or(&out.words[idx], v, __ATOMIC_RELAXED);
                        else out.words[idx] |= v;
                    };
                    for (uint64_t j = 0; j < n_src; j++) {
                        const uint64_t v = b.words[j];
                        put(dw + j, sh ? (v >> sh) : v);
                        if (sh) put(dw + j + 1, v << (64 - sh));
                    }
                    const uint64_t r0 = read_at[i - first];
                    for (uint64_t r = 0; r < b.n_reads(); r++) out.offsets[r0 + r + 1] = at + b.offsets[r + 1];
                }
            });
        for (auto &th : mergers) th.join();
        const double t_m = tnow();
        sink(out);
        if (dbg) fprintf(stderr, "[ingest] merged %zu pieces (%llu reads) in %.3f s, sink %.3f s\n", last - first, (unsigned long long)reads, t_m - t_m0, tnow() - t_m);
        first = last;
    }
    if (dbg) fprintf(stderr, "[ingest] %zu threads: map+cut+sniff %.3f s, parse %.3f s, merge+sink %.3f s\n", n_threads, t_cut - t_begin, t_parse - t_cut, tnow() - t_parse);
    return true;
}

}  // namespace

namespace {
// itmo!/io/ReadersUtils.java:27-53 detectFileFormat on a lower-cased file name without its compression suffix
void reads_format_of(const std::string &name, bool *binq, bool *fastq, bool *fasta)
{
    *binq = ends_with(name, ".binq");
    *fastq = ends_with(name, ".fastq") || ends_with(name, ".fq");
    *fasta = ends_with(name, ".fasta") || ends_with(name, ".fa") || ends_with(name, ".fn") || ends_with(name, ".fna");
}
}  // namespace

PlainReadsFile::~PlainReadsFile()
{
    if (p && n) munmap(const_cast<char *>(p), n);
    if (fd >= 0) close(fd);
}

bool map_plain_reads(const std::string &path, PlainReadsFile *out)
{
    const size_t slash = path.find_last_of('/');
    const std::string name = lower(slash == std::string::npos ? path : path.substr(slash + 1));
    bool binq, fastq, fasta;
    reads_format_of(name, &binq, &fastq, &fasta);
    if (binq || (!fastq && !fasta)) return false;  // (a .gz / .bz2 name matches neither)
    out->fd = open(path.c_str(), O_RDONLY);
    if (out->fd < 0) return false;
    struct stat st;
    if (fstat(out->fd, &st) != 0 || st.st_size <= 0) return false;
    void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, out->fd, 0);
    if (mp == MAP_FAILED) return false;
    out->p = static_cast<const char *>(mp);
    out->n = (size_t)st.st_size;
    out->fastq = fastq;
    if (fastq) {
        MemLines src(out->p, out->p + out->n);
        int offset = -1;
        try {
            if (!parse_fastq(src, [](const char *, size_t) {}, -1, true, &offset, 1) || offset < 0) return false;
        } catch (const Error &) {
            return false;  // the serial reader meets the same problem and reports it
        }
        out->offset = offset;
    }
    return true;
}

const char *plain_record_start(const PlainReadsFile &f, const char *q)
{
    const char *begin = f.p, *end = f.p + f.n;
    auto line_end = [&](const char *s) { const char *nl = static_cast<const char *>(memchr(s, '\n', (size_t)(end - s))); return nl ? nl : end; };
    auto next_line = [&](const char *s) { const char *e = line_end(s); return e < end ? e + 1 : end; };
    auto len_of = [&](const char *s) { const char *e = line_end(s); size_t l = (size_t)(e - s); if (l && s[l - 1] == '\r') l--; return l; };
    const char *l = begin;
    if (q > begin) {
        const char *nl = static_cast<const char *>(memchr(q - 1, '\n', (size_t)(end - (q - 1))));
        l = nl ? nl + 1 : end;
    }
    for (; l < end; l = next_line(l)) {
        if (!f.fastq) {
            if (*l == '>' || *l == ';') return l;
            continue;
        }
        // an '@' line whose third line starts with '+' and whose second and fourth lines are equally long (a quality
        // line that starts with '@' fails the '+' test: the line two below it holds bases)
        if (*l != '@') continue;
        const char *l2 = next_line(l), *l3 = l2 < end ? next_line(l2) : end;
        if (l3 >= end || *l3 != '+') continue;
        const char *l4 = next_line(l3);
        if (l4 < end && len_of(l2) == len_of(l4)) return l;
    }
    return end;
}

uint64_t parse_plain_range(const PlainReadsFile &f, const char *b, const char *e, size_t max_reads, const std::function<void(PackedBatch &)> &sink)
{
    uint64_t delivered = 0;
    PackedBatch batch;
    batch.clear();
    auto emit = [&](const char *s, size_t n) {
        batch.add_read(s, n);
        delivered++;
        if (batch.n_reads() >= max_reads) {
            batch.finish();
            sink(batch);
            batch.clear();
        }
    };
    MemLines src(b, e);
    if (f.fastq) parse_fastq(src, emit, f.offset, false);
    else parse_fasta(src, emit);
    if (batch.n_reads() > 0) {
        batch.finish();
        sink(batch);
    }
    return delivered;
}

uint64_t load_reads_file(const std::string &path, size_t max_reads, const std::function<void(PackedBatch &)> &sink)
{
    // itmo!/io/ReadersUtils.java:27-53 detectFileFormat: a .gz and then a .bz2 suffix come off, then the format
    const size_t slash = path.find_last_of('/');
    const std::string full_name = lower(slash == std::string::npos ? path : path.substr(slash + 1));
    std::string name = full_name, suffix;
    Compression comp = COMP_NONE;
    if (ends_with(name, ".gz")) { comp = COMP_GZ; suffix = ".gz"; name.resize(name.size() - 3); }
    if (ends_with(name, ".bz2")) { comp = COMP_BZ2; suffix = ".bz2"; name.resize(name.size() - 4); }
    const bool binq = ends_with(name, ".binq");
    const bool fastq = ends_with(name, ".fastq") || ends_with(name, ".fq");
    const bool fasta = ends_with(name, ".fasta") || ends_with(name, ".fa") || ends_with(name, ".fn") || ends_with(name, ".fna");
    if (!binq && !fastq && !fasta) throw Error("Can't detect file format for file '" + name + "'");
    if (binq && comp != COMP_NONE) throw Error("Illegal format binq" + suffix);  // ReadersUtils.java:210-214

    uint64_t delivered = 0;
    if (comp == COMP_NONE && !binq && load_reads_parallel(path, fastq, max_reads, sink, &delivered)) return delivered;

    // the reference's readers are serial (one synchronized source per file, src/io/ReadsDispatcher.java:34-53)
    PackedBatch batch;
    batch.clear();
    auto emit = [&](const char *s, size_t n) {
        batch.add_read(s, n);
        delivered++;
        if (batch.n_reads() >= max_reads) {
            batch.finish();
            sink(batch);
            batch.clear();
        }
    };
    if (binq) {
        ByteSource src(path, COMP_NONE);
        parse_binq(src, emit, full_name);
    } else {
        LineSource src(path, comp);
        if (fasta) parse_fasta(src, emit);
        else parse_fastq(src, emit, -1, false);
    }
    if (batch.n_reads() > 0) {
        batch.finish();
        sink(batch);
    }
    return delivered;
}

// ------------------------------------------------------------------------------------------ Environment

// ---- packed k-mers

kmer_t pack_kmer128(const std::string &s)
{
    uint64_t hi, lo;
    pack_kmer(s, &hi, &lo);
    return ((kmer_t)hi << 64) | lo;
}

static inline uint64_t reverse_pairs64(uint64_t x)
{
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
}

kmer_t reverse_complement128(kmer_t v, int k)
{
    // complement is 3 - code (A0<->T3, G1<->C2); then the 2-bit groups in reverse order
    const kmer_t c = ~v;
    const kmer_t r = ((kmer_t)reverse_pairs64((uint64_t)c) << 64) | reverse_pairs64((uint64_t)(c >> 64));
    return r >> (128 - 2 * k);
}

// ASCII order is A < C < G < T, the codes are A0 G1 C2 T3: swapping the two bits of every code gives ranks
// whose numeric order is String.compareTo's
static inline kmer_t ascii_rank(kmer_t v)
{
    const kmer_t m = ((kmer_t)0x5555555555555555ull << 64) | 0x5555555555555555ull;
    return ((v & m) << 1) | ((v >> 1) & m);
}

kmer_t normalize128(kmer_t v, int k)
{
    const kmer_t rc = reverse_complement128(v, k);
    return ascii_rank(v) < ascii_rank(rc) ? v : rc;
}

static inline kmer_t kmer_mask(int bases) { return bases >= 64 ? ~(kmer_t)0 : (((kmer_t)1 << (2 * bases)) - 1); }

// ---- JavaKmerMap

JavaKmerMap::JavaKmerMap(int k) : k_(k), head_(16, NIL), tail_(16, NIL), tree_([this](uint32_t x, uint32_t p) { return tree_dir(x, p); }) {}

int JavaKmerMap::tree_dir(uint32_t x, uint32_t p) const
{   // TreeNode order: the (signed) spread hash, then String.compareTo on the characters the packed keys stand for
    const int32_t h = (int32_t)entries_[x].hash, ph = (int32_t)entries_[p].hash;
    if (ph > h) return -1;
    if (ph < h) return 1;
    return unpack_kmer128(entries_[x].key, k_) < unpack_kmer128(entries_[p].key, k_) ? -1 : 1;
}

void JavaKmerMap::relink(size_t b, const std::vector<uint32_t> &chain)
{
    head_[b] = chain.empty() ? NIL : chain.front();
    tail_[b] = chain.empty() ? NIL : chain.back();
    for (size_t i = 0; i < chain.size(); i++) entries_[chain[i]].next = i + 1 < chain.size() ? chain[i + 1] : NIL;
}

uint32_t JavaKmerMap::hash_of(kmer_t key) const
{
    uint32_t h = 0;  // String.hashCode() of the k characters, then HashMap.hash()'s spread
    for (int i = k_ - 1; i >= 0; i--) h = 31u * h + (uint32_t)(unsigned char)"AGCT"[(unsigned)(key >> (2 * i)) & 3u];
    return h ^ (h >> 16);
}

int JavaKmerMap::find_entry(kmer_t key) const
{
    const uint32_t h = hash_of(key);
    for (uint32_t e = head_[h & (cap_ - 1)]; e != NIL; e = entries_[e].next)
        if (entries_[e].key == key) return (int)e;
    return -1;
}

int JavaKmerMap::put(kmer_t key, int value)
{
    const uint32_t h = hash_of(key);
    const size_t b = h & (cap_ - 1);
    size_t len = 0;
    for (uint32_t e = head_[b]; e != NIL; e = entries_[e].next, len++)
        if (entries_[e].key == key) {
            entries_[e].value = value;
            return (int)e;
        }
    const uint32_t e = (uint32_t)entries_.size();
    entries_.push_back(Entry{key, value, h, NIL});
    auto tb = tree_bins_.find(b);
    if (tb != tree_bins_.end()) {  // a treeified bin: putTreeVal decides where the node goes
        tree_.put(tb->second, e);
        relink(b, tb->second);
    } else {
        if (tail_[b] == NIL) head_[b] = e; else entries_[tail_[b]].next = e;
        tail_[b] = e;
        if (len + 1 >= 9) {  // a 9th node: treeifyBin (or a resize while the table is smaller than 64)
            if (cap_ >= 64) {
                std::vector<uint32_t> chain;
                for (uint32_t x = head_[b]; x != NIL; x = entries_[x].next) chain.push_back(x);
                tree_.treeify(chain);
                relink(b, chain);
                tree_bins_.emplace(b, std::move(chain));
                n_treeified_++;
            } else {
                resize();
            }
        }
    }
    size_++;
    if ((double)size_ > 0.75 * (double)cap_) resize();
    return (int)e;
}

void JavaKmerMap::resize()
{
    const size_t ncap = cap_ * 2;
    std::vector<uint32_t> nh(ncap, NIL), nt(ncap, NIL);
    for (uint32_t h : head_)
        for (uint32_t e = h, nx; e != NIL; e = nx) {  // lo/hi split keeps relative order
            nx = entries_[e].next;
            const size_t b = entries_[e].hash & (ncap - 1);
            entries_[e].next = NIL;
            if (nt[b] == NIL) nh[b] = e; else entries_[nt[b]].next = e;
            nt[b] = e;
        }
    const size_t ocap = cap_;
    cap_ = ncap;
    head_.swap(nh);
    tail_.swap(nt);
    // TreeNode.split for the bins that were trees: halves of <= 6 nodes are plain lists again (in chain order, as linked
    // above); the others stay trees, rebuilt (the root moves to the front) only when the bin really split
    std::unordered_map<size_t, std::vector<uint32_t>> old;
    old.swap(tree_bins_);
    for (auto &kv : old) {
        std::vector<uint32_t> lo, hi;
        for (uint32_t e : kv.second) ((entries_[e].hash & ocap) ? hi : lo).push_back(e);
        const bool both = !lo.empty() && !hi.empty();
        for (int side = 0; side < 2; side++) {
            std::vector<uint32_t> &half = side ? hi : lo;
            if (half.empty()) continue;
            const size_t at = kv.first + (side ? ocap : 0);
            if (half.size() <= 6) { tree_.forget(half); continue; }
            if (both) { tree_.treeify(half); relink(at, half); }
            tree_bins_.emplace(at, std::move(half));
        }
    }
}

void JavaKmerMap::remove(kmer_t key, bool movable)
{   // (movable: HashMap.remove(key) passes true, an iterator's remove -- runTrimPaths' retainAll -- false)
    const size_t b = hash_of(key) & (cap_ - 1);
    auto tb = tree_bins_.find(b);
    if (tb != tree_bins_.end()) {  // TreeNode.removeTreeNode
        std::vector<uint32_t> &chain = tb->second;
        for (size_t i = 0; i < chain.size(); i++)
            if (entries_[chain[i]].key == key) {
                const bool plain = tree_.remove(chain, chain[i], movable);
                relink(b, chain);
                if (plain) { tree_.forget(chain); tree_bins_.erase(tb); }
                size_--;
                return;
            }
        return;
    }
    uint32_t prev = NIL;
    for (uint32_t e = head_[b]; e != NIL; prev = e, e = entries_[e].next)
        if (entries_[e].key == key) {
            if (prev == NIL) head_[b] = entries_[e].next; else entries_[prev].next = entries_[e].next;
            if (tail_[b] == e) tail_[b] = prev;
            size_--;
            return;
        }
}

// ---- Environment

Environment::Environment(int k, std::vector<std::string> gene_sequences) : k_(k), genes_(std::move(gene_sequences)), subgraph_(k)
{
    if (k < 1 || k > 63) throw Error("Environment: k must be in 1..63");
    // isGeneNode (:313-320) is `gene.contains(seq) || gene.contains(rc)` on k-long node labels: a lookup in the
    // set of the genes' k-windows.  Labels are upper-case ACGT, so windows holding anything else never match.
    for (const std::string &g : genes_) {
        kmer_t v = 0;
        int valid = 0;
        for (char c : g) {
            int code;
            switch (c) {
            case 'A': code = 0; break;
            case 'G': code = 1; break;
            case 'C': code = 2; break;
            case 'T': code = 3; break;
            default: code = -1;
            }
            if (code < 0) { valid = 0; continue; }
            v = ((v << 2) | (unsigned)code) & kmer_mask(k_);
            if (++valid >= k_) gene_kmers_.push_back(v);
        }
    }
    std::sort(gene_kmers_.begin(), gene_kmers_.end());
    gene_kmers_.erase(std::unique(gene_kmers_.begin(), gene_kmers_.end()), gene_kmers_.end());
}

void Environment::add_pass(const BfsPass &p, bool trim)
{
    JavaKmerMap d(k_);  // distanceToKmer
    std::vector<int16_t> cov;  // by entry of d
    cov.reserve(p.kmers.size());
    for (size_t i = 0; i < p.kmers.size(); i++) {
        const size_t e = (size_t)d.put(p.kmers[i], p.dist[i]);
        if (e >= cov.size()) cov.resize(e + 1);
        cov[e] = p.cov[i];
    }
    if (trim) {
        // runTrimPaths: reverse BFS from lastKmers through getNeighborsByDir(-dir) inside distanceToKmer
        std::vector<kmer_t> queue;
        std::vector<uint8_t> visited(d.n_entries(), 0);
        for (size_t i = 0; i < p.kmers.size(); i++) {
            const int e = d.find_entry(p.kmers[i]);
            if (p.last[i] && !visited[(size_t)e]) { visited[(size_t)e] = 1; queue.push_back(p.kmers[i]); }
        }
        const int dir = -p.dir;
        const int top = 2 * (k_ - 1);
        for (size_t head = 0; head < queue.size(); head++) {
            const kmer_t kmer = queue[head];
            for (unsigned c = 0; c < 4; c++) {  // A, G, C, T; dir 0 interleaves left and right
                kmer_t nb[2];
                int n = 0;
                if (dir != 1) nb[n++] = ((kmer_t)c << top) | (kmer >> 2);
                if (dir != -1) nb[n++] = ((kmer << 2) & kmer_mask(k_)) | c;
                for (int j = 0; j < n; j++) {
                    const int e = d.find_entry(nb[j]);
                    if (e >= 0 && !visited[(size_t)e]) { visited[(size_t)e] = 1; queue.push_back(nb[j]); }
                }
            }
        }
        for (const kmer_t s : p.kmers) {  // keySet().retainAll(visitedKmers): no reordering
            const int e = d.find_entry(s);
            if (e >= 0 && !visited[(size_t)e]) d.remove(s);
        }
    }
    d.for_each([&](kmer_t kmer, int, int e) { subgraph_.put(normalize128(kmer, k_), cov[(size_t)e]); });
    if (d.treeified()) d_treeified_ = true;  // (a removal from a treeified bin of distanceToKmer: runTrimPaths)
}

static inline void append_uint(std::string &out, unsigned long long v)
{
    char buf[24];
    int n = 0;
    do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) out.push_back(buf[--n]);
}

std::string Environment::graph_txt() const
{
    std::string out;
    out.reserve(subgraph_.size() * (size_t)(k_ + 8));
    subgraph_.for_each([&](kmer_t key, int v, int) {
        for (int i = k_ - 1; i >= 0; i--) out.push_back("AGCT"[(unsigned)(key >> (2 * i)) & 3u]);
        out.push_back(' ');
        if (v < 0) { out.push_back('-'); append_uint(out, (unsigned long long)(-(long long)v)); }
        else append_uint(out, (unsigned long long)v);
        out.push_back('\n');
    });
    return out;
}

void Environment::merge_nodes(int first_plus, int second_minus)
{
    const int first_minus = nodes_[first_plus].rc, second_plus = nodes_[second_minus].rc;
    auto merge_labels = [&](const std::string &a, const std::string &b) {
        if (a.compare(a.size() - (size_t)(k_ - 1), (size_t)(k_ - 1), b, 0, (size_t)(k_ - 1)) != 0)
            throw Error("Labels should be merged, but can not: " + a + " and " + b);
        return a + b.substr((size_t)(k_ - 1));
    };
    const std::string new_seq = merge_labels(nodes_[second_plus].sequence, nodes_[first_plus].sequence);
    const std::string new_seq_rc = merge_labels(nodes_[first_minus].sequence, nodes_[second_minus].sequence);
    nodes_[second_plus].sequence = new_seq;
    nodes_[first_minus].sequence = new_seq_rc;
    nodes_[second_plus].rc = first_minus;
    nodes_[first_minus].rc = second_plus;
    nodes_[first_plus].deleted = true;
    nodes_[second_minus].deleted = true;
}

void Environment::create_picture()
{
    // initializeStructures
    nodes_.clear();
    nodes_.reserve(2 * subgraph_.size());
    std::vector<kmer_t> packed;  // by node id
    packed.reserve(2 * subgraph_.size());
    subgraph_.for_each([&](kmer_t seq, int, int) {
        const kmer_t rc = reverse_complement128(seq, k_);
        const bool g = std::binary_search(gene_kmers_.begin(), gene_kmers_.end(), seq) ||
                       std::binary_search(gene_kmers_.begin(), gene_kmers_.end(), rc);
        const int id = (int)nodes_.size();
        nodes_.push_back(Node{unpack_kmer128(seq, k_), id, g, false, id + 1, {}});
        nodes_.push_back(Node{unpack_kmer128(rc, k_), id + 1, g, false, id, {}});
        packed.push_back(seq);
        packed.push_back(rc);
    });
    // node ids by (k-1)-prefix, in node order: an open-addressing table of chains threaded through next[]
    const size_t n = nodes_.size();
    size_t cap = 16;
    while (cap < 2 * n) cap *= 2;
    struct Slot { kmer_t key; int head, tail; };
    std::vector<Slot> slots(cap, Slot{0, -1, -1});
    std::vector<int> next(n, -1);
    auto slot_of = [&](kmer_t key) {
        uint64_t h = (uint64_t)key ^ ((uint64_t)(key >> 64) * 0x9E3779B97F4A7C15ull);
        h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
        size_t i = (size_t)h & (cap - 1);
        while (slots[i].head >= 0 && slots[i].key != key) i = (i + 1) & (cap - 1);
        return i;
    };
    for (size_t i = 0; i < n; i++) {
        Slot &sl = slots[slot_of(packed[i] >> 2)];
        if (sl.head < 0) { sl.key = packed[i] >> 2; sl.head = (int)i; } else next[(size_t)sl.tail] = (int)i;
        sl.tail = (int)i;
    }
    const kmer_t suffix_mask = kmer_mask(k_ - 1);
    for (size_t i = 0; i < n; i++) {
        auto &dst = nodes_[(size_t)nodes_[i].rc].neighbors;
        for (int j = slots[slot_of(packed[i] & suffix_mask)].head; j >= 0; j = next[(size_t)j]) dst.push_back(j);
    }
    // doMerge
    for (;;) {
        bool acted = false;
        for (size_t i = 0; i < nodes_.size(); i++) {
            if (!nodes_[i].deleted && nodes_[i].neighbors.size() == 1) {
                const int other = nodes_[i].neighbors[0];
                if (nodes_[(size_t)other].neighbors.size() != 1 || nodes_[i].is_gene != nodes_[(size_t)other].is_gene)
                    continue;
                merge_nodes((int)i, other);
                acted = true;
            }
        }
        if (!acted) break;
    }
}

std::string Environment::node_id(const Node &n) const
{
    return std::to_string(std::min(nodes_[(size_t)n.rc].id, n.id) + 1) + (n.is_gene ? "_start" : "");
}

std::string Environment::seqs_fasta(int chunk_length) const
{
    std::string out;
    for (const Node &n : nodes_) {
        const Node &rc = nodes_[(size_t)n.rc];
        if (n.deleted || !(n.id < rc.id) || (int)n.sequence.size() < chunk_length) continue;
        std::set<int> ids;
        for (int j : n.neighbors) ids.insert(std::min(nodes_[(size_t)j].id, nodes_[(size_t)nodes_[(size_t)j].rc].id) + 1);
        for (int j : rc.neighbors) ids.insert(std::min(nodes_[(size_t)j].id, nodes_[(size_t)nodes_[(size_t)j].rc].id) + 1);
        ids.erase(std::min(n.id, rc.id) + 1);
        out += "> Id" + node_id(n) + " Length:" + std::to_string(n.sequence.size()) + " Neighbors:[";
        bool first = true;
        for (int x : ids) {
            if (!first) out += ", ";
            out += std::to_string(x);
            first = false;
        }
        out += "]\n" + n.sequence + "\n";
    }
    return out;
}

std::string Environment::graph_gfa() const
{
    std::string out;
    for (const Node &n : nodes_) {
        if (n.deleted || n.sequence.compare(nodes_[(size_t)n.rc].sequence) > 0) continue;
        long long coverage = 0;
        const std::string &s = n.sequence;
        kmer_t fw = 0, rc = 0;  // the window and its reverse complement, rolled along the label
        int last = 0;
        for (size_t i = 0; i < s.size(); i++) {
            const unsigned code = (unsigned)code_of(s[i]);
            fw = ((fw << 2) | code) & kmer_mask(k_);
            rc = (rc >> 2) | ((kmer_t)(3u - code) << (2 * (k_ - 1)));
            if (i + 1 < (size_t)k_) continue;
            const int e = subgraph_.find_entry(ascii_rank(fw) < ascii_rank(rc) ? fw : rc);
            if (e < 0) throw Error("graph_gfa: k-mer of a node label is not in the subgraph: " + s.substr(i + 1 - (size_t)k_, (size_t)k_));
            last = subgraph_.value_at(e);
            coverage += last;
        }
        coverage += (long long)last * (k_ - 1);
        out += "S\t" + node_id(n) + "\t" + s + "\tLN:i:" + std::to_string(s.size()) + "\tKC:i:" + std::to_string(coverage) +
               (n.is_gene ? "\tCL:Z:GREEN" : "") + "\n";
    }
    for (const Node &i : nodes_) {
        if (i.deleted) continue;
        for (int jx : i.neighbors) {
            const Node &j = nodes_[(size_t)jx];
            if (j.deleted) continue;
            out += "L\t" + node_id(i) + "\t" + (i.sequence.compare(nodes_[(size_t)i.rc].sequence) >= 0 ? "+" : "-") + "\t" +
                   node_id(j) + "\t" + (j.sequence.compare(nodes_[(size_t)j.rc].sequence) <= 0 ? "+" : "-") + "\t" +
                   std::to_string(k_ - 1) + "M\n";
        }
    }
    return out;
}

std::string Environment::tsv_nodes() const
{
    std::string out = "id\tlength\tseq\n";
    for (size_t i = 0; i < nodes_.size(); i++) {
        const Node &n = nodes_[i];
        if (n.deleted || n.sequence.compare(nodes_[(size_t)n.rc].sequence) > 0) continue;
        out += std::to_string(i + 1) + "\t" + std::to_string(n.sequence.size()) + "\t" + n.sequence + "\n";
    }
    return out;
}

std::string Environment::tsv_edges() const
{
    auto nid = [&](const Node &n) {
        const Node &rc = nodes_[(size_t)n.rc];
        return (n.sequence.compare(rc.sequence) <= 0 ? std::to_string(n.id + 1) : "-" + std::to_string(rc.id + 1)) +
               (n.is_gene ? "_start" : "");
    };
    std::string out = "source\ttarget\n";
    for (const Node &i : nodes_) {
        if (i.deleted) continue;
        for (int jx : i.neighbors) {
            const Node &j = nodes_[(size_t)jx];
            if (j.deleted) continue;
            out += nid(nodes_[(size_t)i.rc]) + "\t" + nid(j) + "\tpp\n";
        }
    }
    return out;
}

void Environment::write_all(const std::string &out_prefix, int chunk_length)
{
    const std::string g = graph_txt();
    write_file(out_prefix + "/graph.txt", g);
    write_file(out_prefix + "/env.txt", g);
    create_picture();
    write_file(out_prefix + "/seqs.fasta", seqs_fasta(chunk_length));
    write_file(out_prefix + "/graph.gfa", graph_gfa());
    write_file(out_prefix + "/tsvs/edges.tsv", tsv_edges());
    write_file(out_prefix + "/tsvs/nodes.tsv", tsv_nodes());
}

// ------------------------------------------------------------------------------------------ environment-finder-multi

std::string java_format_6_2f(float x)
{
    std::string body;
    if (x != x) {
        body = "NaN";
    } else if (x == std::numeric_limits<float>::infinity() || x == -std::numeric_limits<float>::infinity()) {
        body = x > 0 ? "Infinity" : "-Infinity";
    } else {
        // the exact decimal value of the float, rounded HALF_UP to two places
        char buf[512];
        snprintf(buf, sizeof buf, "%.160f", (double)(x < 0 ? -x : x));
        std::string d = buf;
        const size_t dot = d.find('.');
        std::string ip = d.substr(0, dot), fp = d.substr(dot + 1);
        bool up = fp[2] >= '5';
        std::string digits = ip + fp.substr(0, 2);
        if (up) {
            int i = (int)digits.size() - 1;
            while (i >= 0 && digits[(size_t)i] == '9') digits[(size_t)i--] = '0';
            if (i >= 0) digits[(size_t)i]++; else digits.insert(digits.begin(), '1');
        }
        body = digits.substr(0, digits.size() - 2) + "." + digits.substr(digits.size() - 2);
        if (std::signbit(x)) body = "-" + body;  // (Java prints -0.00 for negative values that round to zero)
    }
    if (body.size() < 6) body.insert(0, 6 - body.size(), ' ');
    return body;
}

namespace {
// DeBruijnGraphUtils.loadGraph (:13-27)
JavaHashMap load_graph(const std::string &path)
{
    std::vector<std::string> lines;
    if (!read_lines(path, &lines)) throw Error("Couldn't load graph from file " + path);
    JavaHashMap g;
    for (const std::string &line : lines) {
        if (line.empty()) continue;
        const size_t sp = line.find(' ');
        if (sp == std::string::npos || sp + 1 >= line.size()) throw Error("Couldn't load graph from file " + path + ": bad line '" + line + "'");
        size_t end = line.find(' ', sp + 1);
        const std::string num = line.substr(sp + 1, end == std::string::npos ? std::string::npos : end - sp - 1);
        char *stop = nullptr;
        const long v = strtol(num.c_str(), &stop, 10);
        if (num.empty() || *stop) throw Error("Couldn't load graph from file " + path + ": bad depth '" + num + "'");
        g.put(line.substr(0, sp), (int)v);
    }
    return g;
}

struct MultiNode {  // src/algo/MultiNode.java:9-29 (rc and neighbours as indices into the node array)
    std::string sequence;
    int id;
    bool is_gene, deleted = false;
    int rc;
    std::vector<int> neighbors;
    std::set<int> graphs;
};
}  // namespace

MultiResult environment_finder_multi(const std::vector<std::string> &env_paths, const std::string &seq_path, int gene_id)
{
    MultiResult R;
    std::vector<JavaHashMap> graphs;
    for (const std::string &p : env_paths) graphs.push_back(load_graph(p));
    if (graphs.empty()) throw Error("Zero environments given");
    if (graphs.size() > 256) R.log.push_back("WARN Found more than 256 environments. Grayscale graph may be not accurate.");
    int k = -1;
    graphs[0].for_each([&](const std::string &kmer, int) { if (k < 0) k = (int)kmer.size(); });
    if (k < 0) throw Error("The first environment is empty");  // (the reference fails with NoSuchElementException)
    for (const JavaHashMap &g : graphs)
        g.for_each([&](const std::string &kmer, int) {
            if ((int)kmer.size() != k)
                throw Error("K-mers of different lengths encountered: " + std::to_string(k) + " and " + std::to_string(kmer.size()));
        });
    SeedFile sf;
    try {
        sf = read_seed_fasta(seq_path);
    } catch (const Error &) {
        throw Error("Could not load sequence file");
    }
    if (gene_id < 1 || (size_t)gene_id > sf.dnas.size() || (size_t)gene_id > sf.comments.size())
        throw Error("--geneid " + std::to_string(gene_id) + " is outside the sequences of " + seq_path);
    const std::string sequence = sf.dnas[(size_t)gene_id - 1], comment = sf.comments[(size_t)gene_id - 1];
    R.log.push_back("INFO Combining environments for sequence " +
                    ((int)sequence.size() >= 2 * k ? sequence.substr(0, (size_t)k) + "..." + sequence.substr(sequence.size() - (size_t)k) +
                                                         " (length=" + std::to_string(sequence.size()) + ")"
                                                   : sequence));

    // initializeStructures (MultiSequenceCalculator.java:51-100)
    JavaHashMap by_kmer;  // value: node index, -1 = null
    for (const JavaHashMap &g : graphs)
        g.for_each([&](const std::string &kmer, int) {
            by_kmer.put(kmer, -1);
            by_kmer.put(reverse_complement(kmer), -1);
        });
    const size_t size = by_kmer.size();
    std::vector<MultiNode> nodes;
    nodes.reserve(size);
    {
        std::vector<std::string> order;
        by_kmer.for_each([&](const std::string &kmer, int) { order.push_back(kmer); });
        for (const std::string &kmer : order) {
            const std::string rc = reverse_complement(kmer);
            if (kmer.compare(rc) > 0) continue;
            if (nodes.size() + 2 > size)
                throw Error("palindromic k-mer " + kmer + ": the reference fails here (ArrayIndexOutOfBoundsException)");
            const bool is_gene = sequence.find(kmer) != std::string::npos || sequence.find(rc) != std::string::npos;
            const int a = (int)nodes.size();
            MultiNode na, nb;
            na.sequence = kmer; na.id = a; na.is_gene = is_gene; na.rc = a + 1;
            nb.sequence = rc; nb.id = a + 1; nb.is_gene = is_gene; nb.rc = a;
            nodes.push_back(na);
            nodes.push_back(nb);
            by_kmer.put(kmer, a);
            by_kmer.put(rc, a + 1);
        }
    }
    for (size_t i = 0; i < graphs.size(); i++)
        graphs[i].for_each([&](const std::string &kmer, int) {
            const int n = by_kmer.get(kmer);
            nodes[(size_t)n].graphs.insert((int)i);
            nodes[(size_t)nodes[(size_t)n].rc].graphs.insert((int)i);
        });
    for (size_t i = 0; i < nodes.size(); i++)
        for (const char c : {'A', 'G', 'C', 'T'}) {
            int nb;
            if (by_kmer.find(nodes[i].sequence.substr(1) + c, &nb) && nb >= 0) nodes[(size_t)nodes[i].rc].neighbors.push_back(nb);
        }

    // doMerge (:102-139)
    auto merge_labels = [&](const std::string &a, const std::string &b) {
        if (a.substr(a.size() - (size_t)(k - 1)) != b.substr(0, (size_t)(k - 1)))
            throw Error("Labels should be merged, but can not: " + a + " and " + b);
        return a + b.substr((size_t)(k - 1));
    };
    for (;;) {
        bool acted = false;
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i].deleted || nodes[i].neighbors.size() != 1) continue;
            const size_t o = (size_t)nodes[i].neighbors[0];
            if (nodes[o].neighbors.size() != 1 || nodes[i].is_gene != nodes[o].is_gene || nodes[i].graphs != nodes[o].graphs) continue;
            const size_t first_minus = (size_t)nodes[i].rc, second_plus = (size_t)nodes[o].rc;
            const std::string new_seq = merge_labels(nodes[second_plus].sequence, nodes[i].sequence);
            const std::string new_rc = merge_labels(nodes[first_minus].sequence, nodes[o].sequence);
            nodes[second_plus].sequence = new_seq;
            nodes[first_minus].sequence = new_rc;
            nodes[second_plus].rc = (int)first_minus;
            nodes[first_minus].rc = (int)second_plus;
            nodes[i].deleted = nodes[o].deleted = true;
            acted = true;
        }
        if (!acted) break;
    }
    auto min_id = [&](const MultiNode &n) { return std::min(n.id, nodes[(size_t)n.rc].id) + 1; };
    auto label = [&](const MultiNode &n) { return std::to_string(min_id(n)) + (n.is_gene ? "_start" : ""); };

    // outputNodeSequences (:141-160)
    for (const MultiNode &n : nodes) {
        const MultiNode &rc = nodes[(size_t)n.rc];
        if (n.deleted || !(n.id < rc.id)) continue;
        std::set<int> ids;
        for (int j : n.neighbors) ids.insert(min_id(nodes[(size_t)j]));
        for (int j : rc.neighbors) ids.insert(min_id(nodes[(size_t)j]));
        ids.erase(min_id(n));
        R.seqs_fasta += "> Id" + label(n) + " Length:" + std::to_string(n.sequence.size()) + " Neighbors:[";
        bool first = true;
        for (int x : ids) {
            if (!first) R.seqs_fasta += ", ";
            R.seqs_fasta += std::to_string(x);
            first = false;
        }
        R.seqs_fasta += "]\n" + n.sequence + "\n";
    }

    // GFAWriterMulti (:37-146)
    const size_t G = graphs.size();
    auto color = [&](const MultiNode &n) -> std::string {
        if (n.is_gene) return "#00ff00";
        const size_t s = n.graphs.size();
        if (G == 2) return s == 1 ? "#ff0000" : s == 2 ? "#0000ff" : "#000000";
        if (G == 3) {
            static const char *c3[] = {"#000000", "#ff0000", "#0000ff", "#ff00ff", "#ffff00", "#ffaa00", "#00ffff"};
            return s <= 6 ? c3[s] : "#000000";
        }
        const int v = (int)(256 * s / G);
        char buf[32];
        snprintf(buf, sizeof buf, "#%02X%02X%02X", v, v, v);
        return buf;
    };
    for (const MultiNode &n : nodes) {
        if (n.deleted || !(n.id < nodes[(size_t)n.rc].id)) continue;
        long long coverage = 0;
        for (const JavaHashMap &g : graphs)
            for (size_t i = 0; i + (size_t)k <= n.sequence.size(); i++) {
                int c;
                if (g.find(normalize_dna(n.sequence.substr(i, (size_t)k)), &c)) coverage += c;
            }
        const std::string col = color(n);
        R.graph_gfa += "S\t" + label(n) + "\t" + n.sequence + "\tLN:i:" + std::to_string(n.sequence.size()) + "\tKC:i:" +
                       std::to_string(coverage) + "\tCL:Z:" + col + "\tC2:Z:" + col + "\n";
    }
    for (const MultiNode &a : nodes) {
        if (a.deleted) continue;
        for (int j : a.neighbors) {
            const MultiNode &b = nodes[(size_t)j];
            R.graph_gfa += "L\t" + label(a) + "\t" + (a.id < nodes[(size_t)a.rc].id ? "+" : "-") + "\t" + label(b) + "\t" +
                           (b.id > nodes[(size_t)b.rc].id ? "+" : "-") + "\t" + std::to_string(k - 1) + "M\n";
        }
    }
    R.gene_fasta = ">" + comment + "\n" + sequence + "\n";

    // printProbability (EnvironmentFinderMultiMain.java:104-170): 32-bit sums (wrapping like Java's int), float division
    R.jacard_sym = "The[31mWarning! symmetric <<Jaccard distance>> (1 - AB/AUB):\n\n";
    R.jacard_alt = "The[31mWarning! alternative <<Jaccard distance>> (1 - AB/A):\n\n";
    for (size_t i = 0; i < G; i++) {
        R.jacard_sym += env_paths[i];
        R.jacard_alt += env_paths[i];
        for (size_t j = 0; j < G; j++) {
            uint32_t diff = 0, diff_alt = 0, uni = 0;  // (unsigned: wraps like Java's int without undefined behaviour)
            graphs[i].for_each([&](const std::string &kmer, int v) {
                int w;
                if (!graphs[j].find(kmer, &w)) { diff += (uint32_t)v; diff_alt += (uint32_t)v; uni += (uint32_t)v; }
                else { diff += (uint32_t)std::abs(v - w); diff_alt += (uint32_t)std::abs(v - w); uni += (uint32_t)std::max(v, w); }
            });
            graphs[j].for_each([&](const std::string &kmer, int v) {
                int w;
                if (!graphs[i].find(kmer, &w)) { diff += (uint32_t)v; uni += (uint32_t)v; }
            });
            const int inter = (int)(uni - diff), u = (int)uni, ua = (int)(uni - diff_alt);
            R.jacard_sym += java_format_6_2f(1.0f - (float)inter / (float)u) + " ";
            R.jacard_alt += java_format_6_2f(1.0f - (float)inter / (float)ua) + " ";
        }
        R.jacard_sym += "\n";
        R.jacard_alt += "\n";
    }
    R.log.push_back("INFO Finished processing!");
    return R;
}

void write_multi(const MultiResult &r, const std::string &output_dir)
{
    write_file(output_dir + "/seqs.fasta", r.seqs_fasta);
    write_file(output_dir + "/graph.gfa", r.graph_gfa);
    write_file(output_dir + "/gene.fasta", r.gene_fasta);
    write_file(output_dir + "/Jacard_sym.txt", r.jacard_sym);
    write_file(output_dir + "/Jacard_alt.txt", r.jacard_alt);
}

}  // namespace mch
