// Java side of the JNI shim: the original Java host (Runner / tools.EnvironmentFinderMain) drives the MI355X hot path
// through this class instead of IOUtils.loadReads + BigLong2ShortHashMap + OneSequenceCalculator.runBfs.
// UNVERIFIED: the build image has no JDK (SURVEY.md section 8c); written against include/mcgpu.h (ABI v9).
// Build: see integration/README.md.
package gpu;

public final class McGpu implements AutoCloseable {
    static { System.loadLibrary("mcgpu_jni"); }   // libmcgpu_jni.so, which links libmcgpu.so

    /** include/mcgpu.h MC_KEY_*: chosen as src/tools/EnvironmentFinderMain.java:128 chooses its loader. */
    public static final int KEY_PACKED = 0, KEY_POLY = 1, KEY_FNV1A = 2;

    private long handle;   // mc_ctx*

    /** capacityHint: distinct k-mers expected, 0 when unknown (the table then sizes itself from the first batch). */
    public McGpu(int k, int keyMode, int device, long capacityHint) { handle = create(k, keyMode, device, capacityHint); }

    /** --coverage is known before the reads are loaded: lets the counting keep the number of solid k-mers current. */
    public native void setCoverageHint(int minCov);

    /** 2-bit packed reads, layout of include/mcgpu.h: A0 G1 C2 T3 (itmo DnaTools), first base most significant, 32 bases
     *  per word, reads back to back; readOffsets[i] = first base of read i, readOffsets[nReads] = all bases.
     *  Replaces the addAndBound loop of IOUtils.ReadsLoadWorker.process (src/io/IOUtils.java:201-214). */
    public native void addReadsPacked(long[] words, long[] readOffsets, int nReads);

    /** A whole --reads file (FASTA / FASTQ / .gz / .bz2 / .binq) with the reference's N and quality policy; returns the
     *  number of reads.  Replaces one iteration of IOUtils.loadReads' file loop (src/io/IOUtils.java:217-248). */
    public native long addReadsFile(String path);

    /** "Hashtable size: N kmers" (src/io/IOUtils.java:245). */
    public native long finalizeCounts();

    /** BigLong2ShortHashMap.get, batched: -1 when absent, counts saturate at 32767. */
    public native short[] get(long[] keys);

    /** One job per (seed set, direction): seedHi / seedLo = the seed windows as 2-bit packed k-mers (hi word all zero for
     *  k <= 32), dir = -1, 0, +1 as OneSequenceCalculator.runBfs(dir); maxKmers / maxRadius < 0 = not set.
     *  Results in distanceToKmer insertion order (src/algo/OneSequenceCalculator.java:154-214). */
    public native BfsResult[] bfsBatch(long[][] seedHi, long[][] seedLo, int[] dir, int minCov, long maxKmers, long maxRadius);

    @Override public native void close();

    private static native long create(int k, int keyMode, int device, long capacityHint);

    /** null for a job whose seeds hold no solid k-mer (the reference logs "Could not find any k-mers of the target gene"). */
    public static final class BfsResult {
        public long[] hi, lo;    // oriented k-mers, discovery order
        public int[] dist;       // distanceToKmer values
        public short[] cov;      // reads.get(key) of each
        public byte[] last;      // != 0: in lastKmers
        public long levels, lookups;
    }

    // ---- helpers the call sites use

    /** itmo Dna (2 bits a base, A0 G1 C2 T3) -> packed words; `at` = base position where the read starts. */
    public static void packInto(long[] words, long at, byte[] codes, int len) {
        for (int i = 0; i < len; i++) {
            long p = at + i;
            words[(int) (p >>> 5)] |= ((long) (codes[i] & 3)) << (62 - 2 * (int) (p & 31));
        }
    }

    /** the k-mer string -> (hi, lo) as seed words: first base most significant, right-aligned in 128 bits. */
    public static long[] packKmer(String s) {
        long hi = 0, lo = 0;
        for (int i = 0; i < s.length(); i++) {
            int c;
            switch (s.charAt(i)) { case 'A': case 'a': c = 0; break; case 'G': case 'g': c = 1; break; case 'C': case 'c': c = 2; break; default: c = 3; }
            hi = (hi << 2) | (lo >>> 62);
            lo = (lo << 2) | c;
        }
        return new long[] {hi, lo};
    }

    /** (hi, lo) -> the k-mer string */
    public static String kmerString(long hi, long lo, int k) {
        char[] out = new char[k];
        for (int i = k - 1; i >= 0; i--) {
            out[i] = "AGCT".charAt((int) (lo & 3));
            lo = (lo >>> 2) | (hi << 62);
            hi >>>= 2;
        }
        return new String(out);
    }
}
