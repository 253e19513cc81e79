cd /root/repo
for v in "" t2; do
  echo "== lib=${v:-product}"
  if [ -n "$v" ]; then export MC_LIB=metacherchant_amd/lib/libmcgpu_$v.so; else unset MC_LIB; fi
  timeout -k 10 200 python scripts/count_only.py 100 2>&1 | grep -v amdgpu.ids | tail -3
done
unset MC_LIB
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
MC_LIB=metacherchant_amd/lib/libmcgpu_ul64.so python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
bash scripts/gpu_variants.sh old::MC_P3_V2=0 new::MC_P3_V2=1
