"""Python restatement of the host-side half of MetaCherchant's environment-finder
(everything in OneSequenceCalculator after the BFS, the seed reader, the read
ingest policy and the writers).

TEST INFRASTRUCTURE ONLY -- imported by tests/ and __graft_entry__.smoke(),
never by the product package.  Plain Python loops: meant for environments of
up to a few 10^5 k-mers.

Citations: src/... = /root/reference/src/...; itmo!/x = ru/ifmo/genetics/x in
/root/reference/lib/itmo-assembler-src.jar; JDK = java.util.HashMap of JDK 8
(build.xml:36-37 targets 1.8).
"""
import os

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def reverse_complement(s):
    """itmo!/dna/DnaTools.java:139-145"""
    return "".join(_COMP[c] for c in reversed(s))


def normalize_dna(s):
    """src/utils/StringUtils.java:34-41 (String.compareTo = ASCII order)"""
    rc = reverse_complement(s)
    return s if s < rc else rc


def java_string_hash(s):
    """JDK String.hashCode: h = 31*h + c, 32-bit wrap."""
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h


class _TNode:
    """A TreeNode of a treeified bin: the bin's iteration order is the `next` chain, the red-black tree decides where
    a new node is linked into it (java.util.HashMap.TreeNode, JDK 8)."""
    __slots__ = ("e", "hash", "next", "prev", "parent", "left", "right", "red")

    def __init__(self, e, h):
        self.e, self.hash = e, h
        self.next = self.prev = self.parent = self.left = self.right = None
        self.red = False


def _sint(h):
    return h - (1 << 32) if h & 0x80000000 else h


def _tree_dir(h, key, p):
    """HashMap.TreeNode.treeify / putTreeVal: by (signed int) hash, then String.compareTo."""
    ph = p.hash
    if ph > h:
        return -1
    if ph < h:
        return 1
    pk = p.e[0]
    return -1 if key < pk else 1  # (distinct Strings never compare equal; tieBreakOrder is not reached)


def _rotate_left(root, p):
    r = p.right if p is not None else None
    if r is not None:
        rl = p.right = r.left
        if rl is not None:
            rl.parent = p
        pp = r.parent = p.parent
        if pp is None:
            root = r
            r.red = False
        elif pp.left is p:
            pp.left = r
        else:
            pp.right = r
        r.left = p
        p.parent = r
    return root


def _rotate_right(root, p):
    l = p.left if p is not None else None
    if l is not None:
        lr = p.left = l.right
        if lr is not None:
            lr.parent = p
        pp = l.parent = p.parent
        if pp is None:
            root = l
            l.red = False
        elif pp.right is p:
            pp.right = l
        else:
            pp.left = l
        l.right = p
        p.parent = l
    return root


def _balance_insertion(root, x):
    x.red = True
    while True:
        xp = x.parent
        if xp is None:
            x.red = False
            return x
        xpp = xp.parent
        if not xp.red or xpp is None:
            return root
        xppl = xpp.left
        if xp is xppl:
            xppr = xpp.right
            if xppr is not None and xppr.red:
                xppr.red = False
                xp.red = False
                xpp.red = True
                x = xpp
            else:
                if x is xp.right:
                    x = xp
                    root = _rotate_left(root, x)
                    xp = x.parent
                    xpp = None if xp is None else xp.parent
                if xp is not None:
                    xp.red = False
                    if xpp is not None:
                        xpp.red = True
                        root = _rotate_right(root, xpp)
        else:
            if xppl is not None and xppl.red:
                xppl.red = False
                xp.red = False
                xpp.red = True
                x = xpp
            else:
                if x is xp.left:
                    x = xp
                    root = _rotate_right(root, x)
                    xp = x.parent
                    xpp = None if xp is None else xp.parent
                if xp is not None:
                    xp.red = False
                    if xpp is not None:
                        xpp.red = True
                        root = _rotate_left(root, xpp)


def _balance_deletion(root, x):
    """HashMap.TreeNode.balanceDeletion (JDK 8)."""
    while True:
        if x is None or x is root:
            return root
        xp = x.parent
        if xp is None:
            x.red = False
            return x
        if x.red:
            x.red = False
            return root
        xpl = xp.left
        if xpl is x:
            xpr = xp.right
            if xpr is not None and xpr.red:
                xpr.red = False
                xp.red = True
                root = _rotate_left(root, xp)
                xp = x.parent
                xpr = None if xp is None else xp.right
            if xpr is None:
                x = xp
            else:
                sl, sr = xpr.left, xpr.right
                if (sr is None or not sr.red) and (sl is None or not sl.red):
                    xpr.red = True
                    x = xp
                else:
                    if sr is None or not sr.red:
                        if sl is not None:
                            sl.red = False
                        xpr.red = True
                        root = _rotate_right(root, xpr)
                        xp = x.parent
                        xpr = None if xp is None else xp.right
                    if xpr is not None:
                        xpr.red = False if xp is None else xp.red
                        sr = xpr.right
                        if sr is not None:
                            sr.red = False
                    if xp is not None:
                        xp.red = False
                        root = _rotate_left(root, xp)
                    x = root
        else:  # symmetric
            if xpl is not None and xpl.red:
                xpl.red = False
                xp.red = True
                root = _rotate_right(root, xp)
                xp = x.parent
                xpl = None if xp is None else xp.left
            if xpl is None:
                x = xp
            else:
                sl, sr = xpl.left, xpl.right
                if (sl is None or not sl.red) and (sr is None or not sr.red):
                    xpl.red = True
                    x = xp
                else:
                    if sl is None or not sl.red:
                        if sr is not None:
                            sr.red = False
                        xpl.red = True
                        root = _rotate_left(root, xpl)
                        xp = x.parent
                        xpl = None if xp is None else xp.left
                    if xpl is not None:
                        xpl.red = False if xp is None else xp.red
                        sl = xpl.left
                        if sl is not None:
                            sl.red = False
                    if xp is not None:
                        xp.red = False
                        root = _rotate_right(root, xp)
                    x = root


class _TreeBin:
    """One treeified bin: `first` heads the next-chain (always the tree's root: moveRootToFront)."""

    def __init__(self, entries):  # treeifyBin: TreeNodes in list order, then treeify
        self.first = None
        tl = None
        for e in entries:
            p = _TNode(e, _sint(e[2]))
            if tl is None:
                self.first = p
            else:
                p.prev = tl
                tl.next = p
            tl = p
        self.treeify()

    @classmethod
    def from_chain(cls, head, retreeify):
        b = cls.__new__(cls)
        b.first = head
        if retreeify:
            b.treeify()
        return b

    def treeify(self):
        root = None
        x = self.first
        while x is not None:
            nxt = x.next
            x.left = x.right = None
            if root is None:
                x.parent = None
                x.red = False
                root = x
            else:
                p = root
                while True:
                    d = _tree_dir(x.hash, x.e[0], p)
                    xp = p
                    p = p.left if d <= 0 else p.right
                    if p is None:
                        x.parent = xp
                        if d <= 0:
                            xp.left = x
                        else:
                            xp.right = x
                        root = _balance_insertion(root, x)
                        break
            x = nxt
        self._move_root_to_front(root)

    def _root(self):
        r = self.first
        while r.parent is not None:
            r = r.parent
        return r

    def _move_root_to_front(self, root):
        first = self.first
        if root is not None and root is not first:
            rn, rp = root.next, root.prev
            if rn is not None:
                rn.prev = rp
            if rp is not None:
                rp.next = rn
            if first is not None:
                first.prev = root
            root.next = first
            root.prev = None
            self.first = root

    def put(self, e):  # putTreeVal for a key known to be absent
        h, key = _sint(e[2]), e[0]
        root = self._root()
        p = root
        while True:
            d = _tree_dir(h, key, p)
            xp = p
            p = p.left if d <= 0 else p.right
            if p is None:
                xpn = xp.next
                x = _TNode(e, h)
                x.next = xpn
                if d <= 0:
                    xp.left = x
                else:
                    xp.right = x
                xp.next = x
                x.parent = x.prev = xp
                if xpn is not None:
                    xpn.prev = x
                self._move_root_to_front(_balance_insertion(root, x))
                return

    def remove(self, e, movable):
        """removeTreeNode (JDK 8) of the node that holds entry e.  Returns the bin's entries as a plain list when the bin
        untreeifies (too small: judged on the tree as it was BEFORE the removal) or is empty, else None.  The next-chain only
        loses the node; with movable (HashMap.remove; an iterator's remove -- retainAll -- passes false) the root moves to the
        front afterwards."""
        p = self.first
        while p.e is not e:
            p = p.next
        first = root = self.first
        succ, pred = p.next, p.prev
        if pred is None:
            self.first = first = succ
        else:
            pred.next = succ
        if succ is not None:
            succ.prev = pred
        if first is None:
            return []
        while root.parent is not None:
            root = root.parent
        rl = root.left
        if root.right is None or rl is None or rl.left is None:
            return list(self.entries())  # untreeify: plain nodes in chain order
        pl, pr = p.left, p.right
        if pl is not None and pr is not None:
            s = pr
            while s.left is not None:  # the successor
                s = s.left
            s.red, p.red = p.red, s.red
            sr = s.right
            pp = p.parent
            if s is pr:  # p was s's direct parent
                p.parent = s
                s.right = p
            else:
                sp = s.parent
                p.parent = sp
                if sp is not None:
                    if s is sp.left:
                        sp.left = p
                    else:
                        sp.right = p
                s.right = pr
                if pr is not None:
                    pr.parent = s
            p.left = None
            p.right = sr
            if sr is not None:
                sr.parent = p
            s.left = pl
            if pl is not None:
                pl.parent = s
            s.parent = pp
            if pp is None:
                root = s
            elif p is pp.left:
                pp.left = s
            else:
                pp.right = s
            replacement = sr if sr is not None else p
        elif pl is not None:
            replacement = pl
        elif pr is not None:
            replacement = pr
        else:
            replacement = p
        if replacement is not p:
            pp = replacement.parent = p.parent
            if pp is None:
                root = replacement
            elif p is pp.left:
                pp.left = replacement
            else:
                pp.right = replacement
            p.left = p.right = p.parent = None
        r = root if p.red else _balance_deletion(root, replacement)
        if replacement is p:  # detach
            pp = p.parent
            p.parent = None
            if pp is not None:
                if p is pp.left:
                    pp.left = None
                elif p is pp.right:
                    pp.right = None
        if movable:
            self._move_root_to_front(r)
        return None

    def entries(self):
        x = self.first
        while x is not None:
            yield x.e
            x = x.next

    def __len__(self):
        return sum(1 for _ in self.entries())


class JavaHashMap:
    """Iteration-order emulation of java.util.HashMap<String, V> (JDK 8).

    put() appends new keys to the tail of their bin and keeps the position of
    existing keys; resize() doubles at ++size > 0.75*cap and splits every bin
    into lo/hi lists preserving relative order; iteration walks bins in index
    order.  A bin whose 9th node arrives while cap >= 64 is treeified (below 64 the
    table is resized instead): its iteration order is the TreeNodes' next-chain --
    treeify() moves the tree's root to the front, putTreeVal() links a new node right
    behind its tree parent, split() keeps the chain order, untreeifies halves of <= 6
    nodes and re-treeifies the others when the bin really split; a removal (runTrimPaths'
    retainAll, through the key set's iterator) is removeTreeNode + balanceDeletion: the chain
    only loses the node, the bin untreeifies when the tree was too small.  `order_unknown`
    is never set any more (kept for callers that ask).
    """

    def __init__(self):
        self.cap = 16
        self.size = 0
        self.bins = [[] for _ in range(16)]  # a list of entries, or a _TreeBin
        self.index = {}
        self.order_unknown = False
        self.n_treeified = 0  # bins that were treeified at some point (tests)

    @property
    def treeified(self):  # (older name: "the order is not guaranteed")
        return self.order_unknown

    @treeified.setter
    def treeified(self, v):
        self.order_unknown = bool(v)

    def put(self, key, val):
        e = self.index.get(key)
        if e is not None:
            e[1] = val
            return
        h = java_string_hash(key)
        h ^= h >> 16
        e = [key, val, h, True]
        self.index[key] = e
        i = h & (self.cap - 1)
        b = self.bins[i]
        if isinstance(b, _TreeBin):
            b.put(e)
        else:
            b.append(e)
            if len(b) >= 9:  # binCount >= TREEIFY_THRESHOLD - 1 with 8 nodes already there
                if self.cap >= 64:
                    self.bins[i] = _TreeBin(b)
                    self.n_treeified += 1
                else:
                    self._resize()  # treeifyBin resizes instead while tab.length < MIN_TREEIFY_CAPACITY
        self.size += 1
        if self.size > 0.75 * self.cap:
            self._resize()

    def _resize(self):
        ocap, ncap = self.cap, self.cap * 2
        nb = [[] for _ in range(ncap)]
        for j, b in enumerate(self.bins):
            if isinstance(b, _TreeBin):  # TreeNode.split
                lo_h = lo_t = hi_h = hi_t = None
                lc = hc = 0
                x = b.first
                while x is not None:
                    nxt = x.next
                    x.next = None
                    if (x.e[2] & ocap) == 0:
                        x.prev = lo_t
                        if lo_t is None:
                            lo_h = x
                        else:
                            lo_t.next = x
                        lo_t = x
                        lc += 1
                    else:
                        x.prev = hi_t
                        if hi_t is None:
                            hi_h = x
                        else:
                            hi_t.next = x
                        hi_t = x
                        hc += 1
                    x = nxt
                for head, cnt, other, at in ((lo_h, lc, hi_h, j), (hi_h, hc, lo_h, j + ocap)):
                    if head is None:
                        continue
                    if cnt <= 6:  # UNTREEIFY_THRESHOLD: back to a plain list, in chain order
                        lst = []
                        x = head
                        while x is not None:
                            lst.append(x.e)
                            x = x.next
                        nb[at] = lst
                    else:  # (re-treeified only when the bin really split; else the tree stands as it is)
                        nb[at] = _TreeBin.from_chain(head, other is not None)
            else:
                for e in b:
                    nb[e[2] & (ncap - 1)].append(e)
        self.cap = ncap
        self.bins = nb

    def get(self, key):
        e = self.index.get(key)
        return None if e is None else e[1]

    def __contains__(self, key):
        return key in self.index

    def __len__(self):
        return self.size

    def remove(self, key, movable=False):
        """HashMap.removeNode.  movable: HashMap.remove(key) passes true; the removals this pipeline makes come from an
        iterator (runTrimPaths: keySet().retainAll -> Iterator.remove -> removeNode(..., movable = false))."""
        e = self.index.pop(key, None)
        if e is not None:
            e[3] = False
            i = e[2] & (self.cap - 1)
            b = self.bins[i]
            if isinstance(b, _TreeBin):  # TreeNode.removeTreeNode
                plain = b.remove(e, movable)
                if plain is not None:
                    self.bins[i] = plain
            else:
                b.remove(e)
            self.size -= 1

    def items(self):
        for b in self.bins:
            for e in (b.entries() if isinstance(b, _TreeBin) else b):
                yield e[0], e[1]

    def keys(self):
        for k, _ in self.items():
            yield k


# --------------------------------------------------------------------------- readers

def dnaq_to_string(s):
    """itmo!/dna/DnaQ.java:21-30 + :227-229: N/n/. -> nuc 0 ('A'); ACGT case-insensitive."""
    out = []
    for ch in s:
        if ch in "Nn.":
            out.append("A")
        elif ch in "ACGTacgt":
            out.append(ch.upper())
        else:
            raise ValueError("Incorrect nucleotide char: \"%s\"" % ch)
    return "".join(out)


def rich_fasta_read(path):
    """src/io/RichFastaReader.java:38-77 -> (dnas as strings, comments)"""
    dnas, comments = [], []
    last_comment = True
    cur_comment, cur_dna = "", ""
    with open(path, "r") as f:
        data = f.read()
    lines = data.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for line in lines:
        if line.endswith("\r"):
            line = line[:-1]
        if line.startswith(">") or line.startswith(";"):
            if not last_comment:
                dnas.append(dnaq_to_string(cur_dna))
                cur_dna, cur_comment = "", ""
            cur_comment += line[1:]
            last_comment = True
        else:
            if last_comment:
                comments.append(cur_comment)
                cur_dna, cur_comment = "", ""
            cur_dna += line
            last_comment = False
    if len(cur_comment) > 0:
        comments.append(cur_comment)
    if len(cur_dna) > 0:
        dnas.append(dnaq_to_string(cur_dna))
    return dnas, comments


def _lines(path):
    # .gz / .bz2: the same readers over a GZIPInputStream (itmo!/io/readers/FastaGZReader.java, FastqGZReader.java)
    # or Hadoop's BZip2Codec stream (FastaBZ2Reader.java:27, FastqBZ2Reader.java)
    import bz2
    import gzip
    low = path.lower()
    with (gzip.open(path, "rt") if low.endswith(".gz") else bz2.open(path, "rt") if low.endswith(".bz2") else open(path, "r")) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.endswith("\r"):
                line = line[:-1]
            yield line


def read_fasta_reads(path):
    """itmo!/io/readers/FastaReader.java:54-104: multi-line records concatenated,
    records containing N/n dropped whole."""
    out = []
    sb = []

    def flush():
        if sb:
            s = "".join(sb)
            if len(s) > 0 and "N" not in s and "n" not in s:
                out.append(s.upper())
        sb.clear()

    for line in _lines(path):
        if line.startswith(">") or line.startswith(";"):
            if len("".join(sb)) > 0:
                flush()
        else:
            sb.append(line)
    flush()
    return out


def read_fastq_reads(path):
    """itmo!/io/readers/FastqReader.java:53-112 + FastaReaderFromXQSourceTrunc.java:61-95 +
    itmo!/io/ReadersUtils.java:57-77: records split at every base with phred < 1 (the bad base
    is dropped, every piece kept); quality offset Illumina+64 unless the first 1000 records hold
    a char < 64, then Sanger+33."""
    recs = []
    it = iter(_lines(path))

    def next_data():
        for s in it:
            if len(s) == 0:
                continue
            if not (s.startswith("@") or s.startswith("+")):
                raise ValueError("Unknown structure of fastq file")
            try:
                return next(it)
            except StopIteration:
                raise ValueError("Unexpected end of file")
        return None

    while True:
        d = next_data()
        if d is None:
            break
        q = next_data()
        if q is None or len(q) != len(d):
            raise ValueError("Bad DnaQ record")
        recs.append((d, q))
    offset = 64
    for d, q in recs[:1000]:
        bad = False
        for ch_d, ch_q in zip(d, q):
            if ch_d in "Nn.":
                continue
            if ord(ch_q) < 64 or ord(ch_q) > 126:
                bad = True
                break
        if bad:
            offset = 33
            break
    out = []
    for d, q in recs:
        piece = []
        for ch_d, ch_q in zip(d, q):
            if ch_d in "Nn.":
                ph = 0
            else:
                ph = ord(ch_q) - offset
                if ph < 0 or ord(ch_q) > 126:
                    raise ValueError("Invalid quality code char")
                ph &= 63  # itmo!/dna/DnaQBuilder.java:32-35 keeps the phred in 6 bits of a byte: 64 reads back as 0
            if ph < 1:
                out.append("".join(piece))
                piece = []
            else:
                piece.append(ch_d.upper())
        out.append("".join(piece))
    return [p for p in out if p]


def read_binq_reads(path):
    """itmo!/io/readers/BinqReader.java:52-86 (records: 4-byte big-endian length, then bytes phred << 2 | nuc;
    0xFF bytes in front of a record are padding) + FastaReaderFromXQSourceTrunc.java:61-95 (pieces)."""
    data = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        while pos < len(data) and data[pos] == 255:
            pos += 1
        if pos >= len(data):
            break
        if pos + 4 > len(data):
            raise ValueError("Unexpected end of file")
        n = int.from_bytes(data[pos:pos + 4], "big")
        pos += 4
        if pos + n > len(data):
            raise ValueError("Unexpected end of file")
        piece = []
        for v in data[pos:pos + n]:
            if (v >> 2) < 1:
                out.append("".join(piece))
                piece = []
            else:
                piece.append("AGCT"[v & 3])
        out.append("".join(piece))
        pos += n
    return [p for p in out if p]


# --------------------------------------------------------------------------- calculator

class SingleNode:
    """src/algo/SingleNode.java:6-56"""
    __slots__ = ("sequence", "id", "is_gene", "deleted", "rc", "neighbors", "color")

    def __init__(self, sequence, id_, color, is_gene):
        self.sequence = sequence
        self.id = id_
        self.is_gene = is_gene
        self.color = color
        self.deleted = False
        self.rc = None
        self.neighbors = []


class Environment:
    """State of one OneSequenceCalculator after buildEnvironment()."""

    def __init__(self, k, seeds, merge):
        self.k = k
        self.seeds = seeds  # list of strings (--seq sequences only; Hi-C seeds excluded from isGeneNode)
        self.merge = merge
        self.subgraph = JavaHashMap()
        self.nodes = None

    # src/algo/OneSequenceCalculator.java:217-219 + :146-148
    def add_pass(self, kmers, dists, covs, kept=None):
        """One runBfs pass: oriented k-mer strings in discovery order."""
        d = JavaHashMap()
        for s, dist in zip(kmers, dists):
            d.put(s, dist)
        if kept is not None:  # :261 keySet().retainAll(visitedKmers)
            for s, kp in zip(kmers, kept):
                if not kp:
                    d.remove(s)
        cov = dict(zip(kmers, covs))
        for s in d.keys():
            self.subgraph.put(normalize_dna(s), int(cov[s]))
        if d.treeified:
            self.subgraph.treeified = True

    # :297-310
    def graph_txt(self):
        return "".join("%s %d\n" % (s, c) for s, c in self.subgraph.items())

    # :421-432
    def _is_gene(self, seq, rc):
        for s in self.seeds:
            if seq in s or rc in s:
                return True
        return False

    # :387-419
    def initialize_structures(self):
        k = self.k
        nodes = []
        for seq, _ in self.subgraph.items():
            rc = reverse_complement(seq)
            g = self._is_gene(seq, rc)
            a = SingleNode(seq, len(nodes), "GREEN" if g else None, g)
            b = SingleNode(rc, len(nodes) + 1, "GREEN" if g else None, g)
            a.rc, b.rc = b, a
            nodes += [a, b]
        by_prefix = {}
        for n in nodes:
            by_prefix.setdefault(n.sequence[: k - 1], []).append(n)
        for n in nodes:
            lst = by_prefix.get(n.sequence[1:])
            if lst is not None:
                n.rc.neighbors.extend(lst)
        self.nodes = nodes

    # :312-324 + :453-462
    def _merge_nodes(self, first_plus, second_minus):
        k = self.k
        first_minus, second_plus = first_plus.rc, second_minus.rc

        def merge_labels(a, b):
            if a[len(a) - (k - 1):] != b[: k - 1]:
                raise AssertionError("Labels should be merged, but can not: %s and %s" % (a, b))
            return a + b[k - 1:]

        new_seq = merge_labels(second_plus.sequence, first_plus.sequence)
        new_seq_rc = merge_labels(first_minus.sequence, second_minus.sequence)
        second_plus.sequence = new_seq
        first_minus.sequence = new_seq_rc
        second_plus.rc = first_minus
        first_minus.rc = second_plus
        first_plus.deleted = True
        second_minus.deleted = True

    # :434-451
    def do_merge(self):
        nodes = self.nodes
        while True:
            acted = False
            for n in nodes:
                if not n.deleted and len(n.neighbors) == 1:
                    other = n.neighbors[0]
                    if len(other.neighbors) != 1 or n.is_gene != other.is_gene:
                        continue
                    self._merge_nodes(n, other)
                    acted = True
            if not acted:
                break

    @staticmethod
    def _node_id(a):  # :464-466 / GFAWriter.java:84-86
        return "%d%s" % (min(a.rc.id, a.id) + 1, "_start" if a.is_gene else "")

    # :354-385
    def seqs_fasta(self, chunk_length):
        out = []
        for n in self.nodes:
            if not n.deleted and n.id < n.rc.id and len(n.sequence) >= chunk_length:
                ids = set()
                for nb in n.neighbors:
                    ids.add(min(nb.id, nb.rc.id) + 1)
                for nb in n.rc.neighbors:
                    ids.add(min(nb.id, nb.rc.id) + 1)
                ids.discard(min(n.id, n.rc.id) + 1)
                out.append("> Id%s Length:%d Neighbors:[%s]\n%s\n" % (
                    self._node_id(n), len(n.sequence), ", ".join(str(x) for x in sorted(ids)), n.sequence))
        return "".join(out)

    # src/io/writers/GFAWriter.java:47-99
    def graph_gfa(self):
        k = self.k
        out = []
        for n in self.nodes:
            if not n.deleted and n.sequence <= n.rc.sequence:
                cov = 0
                s = n.sequence
                for i in range(len(s) - k + 1):
                    cov += self.subgraph.get(normalize_dna(s[i:i + k]))
                cov += self.subgraph.get(normalize_dna(s[len(s) - k:])) * (k - 1)
                out.append("S\t%s\t%s\tLN:i:%d\tKC:i:%d%s\n" % (
                    self._node_id(n), s, len(s), cov, "" if n.color is None else "\tCL:Z:" + n.color))
        for i in self.nodes:
            if not i.deleted:
                for j in i.neighbors:
                    if not j.deleted:
                        out.append("L\t%s\t%s\t%s\t%s\t%dM\n" % (
                            self._node_id(i), "+" if i.sequence >= i.rc.sequence else "-",
                            self._node_id(j), "+" if j.sequence <= j.rc.sequence else "-", k - 1))
        return "".join(out)

    # src/io/writers/TSVWriter.java:27-79
    def tsv_nodes(self):
        out = ["id\tlength\tseq\n"]
        for i, n in enumerate(self.nodes):
            if not n.deleted and n.sequence <= n.rc.sequence:
                out.append("%d\t%d\t%s\n" % (i + 1, len(n.sequence), n.sequence))
        return "".join(out)

    def tsv_edges(self):
        def nid(n):
            base = str(n.id + 1) if n.sequence <= n.rc.sequence else "-" + str(n.rc.id + 1)
            return base + ("_start" if n.is_gene else "")

        out = ["source\ttarget\n"]
        for i in self.nodes:
            if not i.deleted:
                for j in i.neighbors:
                    if not j.deleted:
                        out.append("%s\t%s\tpp\n" % (nid(i.rc), nid(j)))
        return "".join(out)

    # :326-339 createPicture
    def files(self, chunk_length):
        g = self.graph_txt()
        self.initialize_structures()
        self.do_merge()
        return {
            "graph.txt": g,
            "seqs.fasta": self.seqs_fasta(chunk_length),
            "graph.gfa": self.graph_gfa(),
            "tsvs/nodes.tsv": self.tsv_nodes(),
            "tsvs/edges.tsv": self.tsv_edges(),
        }


def write_files(files, out_dir):
    for name, text in files.items():
        p = os.path.join(out_dir, name)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)


# --------------------------------------------------------------------------- driver

def environment_finder(table, k, mode, seqs, comments, out_dir, coverage=1, max_kmers=None, max_radius=None,
                       bothdirs=False, chunk_length=1, trim=False, merge=False, hic_seqs=None):
    """src/tools/EnvironmentFinderMain.java:185-243 runImpl + OneSequenceCalculator.run (:98-114),
    on an already counted oracle table (oracle.pyoracle.Table).  Returns {output_prefix: files or None}."""
    from . import pyoracle as po

    if max_kmers is None and max_radius is None:
        raise ValueError("At least one of --maxkmers and --maxradius parameters should be set")
    mk = -1 if max_kmers is None else max_kmers
    mr = -1 if max_radius is None else max_radius
    jobs = []
    if not merge:
        for i, s in enumerate(seqs):
            jobs.append((os.path.join(out_dir, comments[i]) + "/", [s], [s]))
    else:
        jobs.append((os.path.join(out_dir, "merged") + "/", list(seqs) + list(hic_seqs or []), list(seqs)))
    results = {}
    for prefix, bfs_seeds, gene_seqs in jobs:
        env = Environment(k, gene_seqs, merge)
        fail = False
        for d in ([0] if bothdirs else [-1, 1]):  # :137-144
            r = po.bfs(table, k, mode, [po.encode(s) for s in bfs_seeds], d, coverage, mk, mr, trim)
            if r is None:
                fail = True
                break
            kmers = [po.kmer_string(h, l, k) for h, l in zip(r["hi"], r["lo"])]
            env.add_pass(kmers, r["dist"], r["cov"], r["kept"] if trim else None)
        if fail:
            results[prefix] = None  # "Could not find any k-mers of the target gene in the input, halting."
            continue
        files = env.files(chunk_length)
        files["env.txt"] = files["graph.txt"]
        write_files(files, prefix)
        results[prefix] = files
    return results


# ---- --tool kmer-counter (src/tools/KmersCounter.java:87-121, src/io/IOUtils.java:39-65,
# itmo!/statistics/QuickQuantitativeStatistics.java:37-72)

def library_name(path):
    """ReadersUtils.readDnaLazy(file).name(): file name without its format extension."""
    name = os.path.basename(path)
    low = name.lower()
    for ext in (".fasta.gz", ".fa.gz", ".fn.gz", ".fna.gz", ".fastq.gz", ".fq.gz", ".fasta.bz2", ".fa.bz2", ".fn.bz2", ".fna.bz2",
                ".fastq.bz2", ".fq.bz2", ".fasta", ".fa", ".fn", ".fna", ".fastq", ".fq", ".binq"):
        if low.endswith(ext):
            return name[:len(name) - len(ext)]
    return name


def kmer_counter_files(table, threshold=0):
    """(records of <name>.kmers.bin as a sorted list of (key, count), text of <name>.stat.txt).
    Record order in the reference is its map's iteration order (unpinned: fastutil hash), so records
    are compared as a set; the statistics file is sorted by frequency and must match byte for byte."""
    keys, counts = table.dump()
    recs = sorted((int(k), int(c)) for k, c in zip(keys, counts) if c > threshold)
    hist = {}
    for c in counts:
        hist[int(c)] = hist.get(int(c), 0) + 1
    text = "# k-mer frequency\tnumber of such k-mers\n" + "".join("%d\t%d\n" % (v, hist[v]) for v in sorted(hist)) + "\n"
    return recs, text


def read_kmers_bin(path):
    """10-byte records: big-endian int64 key, big-endian int16 count (src/io/KmersLoadWorker.java:9-23)."""
    import struct
    data = open(path, "rb").read()
    assert len(data) % 10 == 0
    return sorted(struct.unpack(">qh", data[i:i + 10]) for i in range(0, len(data), 10))


# ---- --tool environment-finder-multi (src/tools/EnvironmentFinderMultiMain.java, src/algo/MultiSequenceCalculator.java,
# src/algo/MultiNode.java, src/io/writers/GFAWriterMulti.java, src/io/graph/DeBruijnGraphUtils.java)

def load_graph(path):
    """DeBruijnGraphUtils.loadGraph (:13-27): "<kmer> <depth>" lines into a java.util.HashMap, in file order."""
    g = JavaHashMap()
    with open(path, "r") as f:
        for line in f.read().split("\n"):
            if line == "" :
                continue  # (BufferedReader.ready() is false at the end of the file; a blank line would crash the reference)
            tokens = line.rstrip("\r").split(" ")
            g.put(tokens[0], int(tokens[1]))
    return g


class MultiNode:
    """src/algo/MultiNode.java:9-29"""
    __slots__ = ("sequence", "id", "is_gene", "deleted", "rc", "neighbors", "graphs")

    def __init__(self, sequence, id_, is_gene):
        self.sequence, self.id, self.is_gene = sequence, id_, is_gene
        self.deleted = False
        self.rc = None
        self.neighbors = []
        self.graphs = set()


def java_format_6_2f(x):
    """String.format("%6.2f", (float) x): the exact decimal value of the float, rounded HALF_UP to two places."""
    import decimal
    import math
    x = float(x)
    if math.isnan(x):
        s = "NaN"
    elif math.isinf(x):
        s = "Infinity" if x > 0 else "-Infinity"
    else:
        d = decimal.Decimal(x).quantize(decimal.Decimal("0.01"), rounding=decimal.ROUND_HALF_UP)
        s = "%.2f" % d if d != 0 else ("-0.00" if math.copysign(1.0, x) < 0 else "0.00")
    return s.rjust(6)


def environment_finder_multi(env_paths, seq_path, out_dir, gene_id=1):
    """Returns ({file name: text}, log lines); writes nothing.  Raises ValueError where the reference fails."""
    import numpy as np
    graphs = [load_graph(p) for p in env_paths]
    if not graphs:
        raise ValueError("Zero environments given")
    log = []
    if len(graphs) > 256:
        log.append("WARN Found more than 256 environments. Grayscale graph may be not accurate.")
    k = len(next(iter(graphs[0].keys())))
    for g in graphs:
        for kmer in g.keys():
            if len(kmer) != k:
                raise ValueError("K-mers of different lengths encountered: %d and %d" % (k, len(kmer)))
    dnas, comments = rich_fasta_read(seq_path)
    sequence, comment = dnas[gene_id - 1], comments[gene_id - 1]
    log.append("INFO Combining environments for sequence " +
               (sequence[:k] + "..." + sequence[len(sequence) - k:] + " (length=%d)" % len(sequence) if len(sequence) >= 2 * k else sequence))

    # initializeStructures (MultiSequenceCalculator.java:51-100)
    by_kmer = JavaHashMap()
    for g in graphs:
        for kmer in g.keys():
            by_kmer.put(kmer, None)
            by_kmer.put(reverse_complement(kmer), None)
    size = len(by_kmer)
    nodes = []
    for kmer in list(by_kmer.keys()):
        rc = reverse_complement(kmer)
        if kmer > rc:
            continue
        if len(nodes) + 2 > size:  # a palindromic k-mer (even k) takes two nodes but one key: nodes[] is too short
            raise ValueError("palindromic k-mer %s: the reference fails here (ArrayIndexOutOfBoundsException)" % kmer)
        is_gene = kmer in sequence or rc in sequence
        a, b = MultiNode(kmer, len(nodes), is_gene), MultiNode(rc, len(nodes) + 1, is_gene)
        a.rc, b.rc = b, a
        nodes += [a, b]
        by_kmer.put(a.sequence, a)
        by_kmer.put(b.sequence, b)
    for i, g in enumerate(graphs):
        for kmer in g.keys():
            n = by_kmer.get(kmer)
            n.graphs.add(i)
            n.rc.graphs.add(i)
    for n in nodes:
        for c in "AGCT":
            nb = by_kmer.get(n.sequence[1:] + c)
            if nb is not None:
                n.rc.neighbors.append(nb)

    # doMerge (:102-122, 124-139)
    def merge_labels(a, b):
        if a[len(a) - (k - 1):] != b[: k - 1]:
            raise AssertionError("Labels should be merged, but can not: %s and %s" % (a, b))
        return a + b[k - 1:]

    while True:
        acted = False
        for n in nodes:
            if not n.deleted and len(n.neighbors) == 1:
                o = n.neighbors[0]
                if len(o.neighbors) == 1 and n.is_gene == o.is_gene and n.graphs == o.graphs:
                    first_minus, second_plus = n.rc, o.rc
                    new_seq = merge_labels(second_plus.sequence, n.sequence)
                    new_rc = merge_labels(first_minus.sequence, o.sequence)
                    second_plus.sequence, first_minus.sequence = new_seq, new_rc
                    second_plus.rc, first_minus.rc = first_minus, second_plus
                    n.deleted = o.deleted = True
                    acted = True
        if not acted:
            break

    def min_id(n):
        return min(n.id, n.rc.id) + 1

    # outputNodeSequences (:141-160)
    seqs = []
    for n in nodes:
        if not n.deleted and n.id < n.rc.id:
            ids = set(min_id(x) for x in n.neighbors) | set(min_id(x) for x in n.rc.neighbors)
            ids.discard(min_id(n))
            seqs.append("> Id%d%s Length:%d Neighbors:[%s]\n%s\n" % (
                min_id(n), "_start" if n.is_gene else "", len(n.sequence), ", ".join(str(x) for x in sorted(ids)), n.sequence))

    # GFAWriterMulti (:37-146)
    def color(n):
        G = len(graphs)
        if n.is_gene:
            return "#00ff00"
        s = len(n.graphs)
        if G == 2:
            return {1: "#ff0000", 2: "#0000ff"}.get(s, "#000000")
        if G == 3:
            return {1: "#ff0000", 2: "#0000ff", 3: "#ff00ff", 4: "#ffff00", 5: "#ffaa00", 6: "#00ffff"}.get(s, "#000000")
        v = 256 * s // G
        return "#%02X%02X%02X" % (v, v, v)

    gfa = []
    for n in nodes:
        if not n.deleted and n.id < n.rc.id:
            cov = 0
            for g in graphs:
                for i in range(len(n.sequence) - k + 1):
                    c = g.get(normalize_dna(n.sequence[i:i + k]))
                    cov += 0 if c is None else c
            col = color(n)
            gfa.append("S\t%d%s\t%s\tLN:i:%d\tKC:i:%d\tCL:Z:%s\tC2:Z:%s\n" % (
                min_id(n), "_start" if n.is_gene else "", n.sequence, len(n.sequence), cov, col, col))
    for a in nodes:
        if not a.deleted:
            for b in a.neighbors:
                gfa.append("L\t%d%s\t%s\t%d%s\t%s\t%dM\n" % (
                    min_id(a), "_start" if a.is_gene else "", "+" if a.id < a.rc.id else "-",
                    min_id(b), "_start" if b.is_gene else "", "+" if b.id > b.rc.id else "-", k - 1))

    # printProbability (EnvironmentFinderMultiMain.java:104-170): 32-bit int sums, float division
    G = len(graphs)
    dicts = [dict(g.items()) for g in graphs]
    sym = ["The[31mWarning! symmetric <<Jaccard distance>> (1 - AB/AUB):\n", "\n"]
    alt = ["The[31mWarning! alternative <<Jaccard distance>> (1 - AB/A):\n", "\n"]
    for i in range(G):
        sym.append(str(env_paths[i]))
        alt.append(str(env_paths[i]))
        for j in range(G):
            F, S = dicts[i], dicts[j]
            diff = diff_alt = union = 0
            for kmer, v in F.items():
                if kmer not in S:
                    diff += v; diff_alt += v; union += v
                else:
                    diff += abs(v - S[kmer]); diff_alt += abs(v - S[kmer]); union += max(v, S[kmer])
            for kmer, v in S.items():
                if kmer not in F:
                    diff += v; union += v
            inter = union - diff
            with np.errstate(divide="ignore", invalid="ignore"):
                sym.append(java_format_6_2f(np.float32(1) - np.float32(inter) / np.float32(union)) + " ")
                alt.append(java_format_6_2f(np.float32(1) - np.float32(inter) / np.float32(union - diff_alt)) + " ")
        sym.append("\n")
        alt.append("\n")
    log.append("INFO Finished processing!")
    files = {"seqs.fasta": "".join(seqs), "graph.gfa": "".join(gfa), "gene.fasta": ">%s\n%s\n" % (comment, sequence),
             "Jacard_sym.txt": "".join(sym), "Jacard_alt.txt": "".join(alt)}
    return files, log
