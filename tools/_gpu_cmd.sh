mkdir -p gpurun_out/num
L=$PWD/exploration-of-potential_amd/ep24
for cfg in "1.0 1.0 640 4" "0.25 0.33 128 2"; do
  EP24_LIB=$L/libep24_r05g.so timeout -k 10 300 python tools/numeric_ab.py run gpurun_out/num/a.pt $cfg 2>&1 | grep -v amdgpu
  EP24_LIB=$L/libep24_r05g.so timeout -k 10 300 python tools/numeric_ab.py run gpurun_out/num/a2.pt $cfg 2>&1 | grep -v amdgpu
  EP24_LIB=$L/libep24.so timeout -k 10 300 python tools/numeric_ab.py run gpurun_out/num/b.pt $cfg 2>&1 | grep -v amdgpu
  echo "== $cfg: r05g library twice (reproducibility), then r05g against HEAD"
  python tools/numeric_ab.py cmp gpurun_out/num/a.pt gpurun_out/num/a2.pt
  python tools/numeric_ab.py cmp gpurun_out/num/b.pt gpurun_out/num/a.pt
done
rm -f gpurun_out/num/*.pt
