// mc_hosttest -- CPU-only driver of the host logic (no GPU, no libmcgpu): takes BFS passes from
// a dump file instead of mc_bfs_batch and writes the environment files, so that tests/ can compare
// the C++ host side with the Python restatement in oracle/host_oracle.py.
//
// dump format (text): k chunk_length trim n_genes / gene strings / n_passes / per pass: dir n /
// n lines "kmer dist cov last".  Also: `mc_hosttest seeds <fasta>` and `mc_hosttest reads <file>`
// print what the seed reader / read ingest deliver.
#include <cstdio>
#include <fstream>
#include <functional>
#include <iostream>

#include "envfinder.h"

using namespace mch;

int main(int argc, char **argv)
{
    try {
        if (argc == 3 && std::string(argv[1]) == "seeds") {
            const SeedFile s = read_seed_fasta(argv[2]);
            printf("%zu %zu\n", s.dnas.size(), s.comments.size());
            for (const auto &d : s.dnas) printf("D %s\n", d.c_str());
            for (const auto &c : s.comments) printf("C %s\n", c.c_str());
            return 0;
        }
        if (argc == 3 && std::string(argv[1]) == "reads") {
            const uint64_t n = load_reads_file(argv[2], 1000, [&](PackedBatch &b) {
                for (uint64_t r = 0; r < b.n_reads(); r++) {
                    std::string s;
                    for (uint64_t p = b.offsets[r]; p < b.offsets[r + 1]; p++)
                        s.push_back("AGCT"[(b.words[p >> 5] >> (62 - 2 * (p & 31))) & 3]);
                    printf("%s\n", s.c_str());
                }
            });
            fprintf(stderr, "%llu reads\n", (unsigned long long)n);
            return 0;
        }
        if (argc == 3 && std::string(argv[1]) == "fmt") {  // String.format("%6.2f", (float) x)
            fputs(java_format_6_2f(strtof(argv[2], nullptr)).c_str(), stdout);
            return 0;
        }
        if (argc == 3 && std::string(argv[1]) == "hashmap") {  // put every k-mer string of the file ("-kmer": remove it as an iterator does, "~kmer": as HashMap.remove does), print both maps' orders
            std::ifstream f(argv[2]);
            if (!f) throw Error("cannot open key file");
            std::vector<std::string> ops;
            for (std::string line; std::getline(f, line);) if (!line.empty()) ops.push_back(line);
            const int k = (int)(ops.empty() ? 1 : ops[0].size() - (ops[0][0] == '-' || ops[0][0] == '~'));
            JavaHashMap hm;
            JavaKmerMap km(k);
            int v = 0;
            for (const std::string &op : ops) {
                if (op[0] == '-' || op[0] == '~') { hm.remove(op.substr(1), op[0] == '~'); km.remove(pack_kmer128(op.substr(1)), op[0] == '~'); }
                else { hm.put(op, v); km.put(pack_kmer128(op), v); v++; }
            }
            printf("S %zu %zu %d\n", hm.size(), hm.bins_treeified(), hm.treeified() ? 1 : 0);
            hm.for_each([](const std::string &key, int val) { printf("s %s %d\n", key.c_str(), val); });
            printf("K %zu %zu %d\n", km.size(), km.bins_treeified(), km.treeified() ? 1 : 0);
            km.for_each([&](kmer_t key, int val, int) { printf("k %s %d\n", unpack_kmer128(key, k).c_str(), val); });
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "multi") {  // multi <out_dir> <seq.fasta> <gene_id> <env>...
            const MultiResult r = environment_finder_multi(std::vector<std::string>(argv + 5, argv + argc), argv[3], atoi(argv[4]));
            write_multi(r, argv[2]);
            for (const auto &l : r.log) printf("%s\n", l.c_str());
            return 0;
        }
        if (argc != 4 || std::string(argv[1]) != "env") {
            fprintf(stderr, "usage: mc_hosttest env <dump> <out_prefix> | seeds <fasta> | reads <file> | hashmap <keys> | multi <out_dir> <seq> <gene_id> <env>...\n");
            return 2;
        }
        std::ifstream f(argv[2]);
        if (!f) throw Error("cannot open dump");
        int k, chunk, trim, n_genes;
        f >> k >> chunk >> trim >> n_genes;
        std::vector<std::string> genes((size_t)n_genes);
        for (auto &g : genes) f >> g;
        int n_passes;
        f >> n_passes;
        Environment env(k, genes);
        for (int p = 0; p < n_passes; p++) {
            BfsPass pass;
            size_t n;
            f >> pass.dir >> n;
            pass.kmers.resize(n); pass.dist.resize(n); pass.cov.resize(n); pass.last.resize(n);
            for (size_t i = 0; i < n; i++) {
                int d, c, l;
                std::string kmer;
                f >> kmer >> d >> c >> l;
                pass.kmers[i] = pack_kmer128(kmer);
                pass.dist[i] = d; pass.cov[i] = (int16_t)c; pass.last[i] = (uint8_t)l;
            }
            env.add_pass(pass, trim != 0);
        }
        env.write_all(argv[3], chunk);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
