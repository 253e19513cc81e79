cd /root/repo
MC_LIB=metacherchant_amd/lib/libmcgpu_p1time.so timeout -k 10 300 python bench.py --config 2 --reads 10000000 --contigs 10 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_p1time.json 2> gpurun_out/b_p1time.err
grep -h "p1 block" gpurun_out/b_p1time.err gpurun_out/b_p1time.json | tail -4
tail -c 600 gpurun_out/b_p1time.json
