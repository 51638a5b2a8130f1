cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz6
for f in fuzz_api fuzz_api_misc fuzz_resident fuzz_streaming fuzz_kernels fuzz_lanczos fuzz_pivchol; do
  timeout 200 python tools/$f.py --minutes 2 > gpurun_out/fuzz6/$f.log 2>&1; echo "$f rc=$? : $(tail -1 gpurun_out/fuzz6/$f.log | cut -c1-200)"
done
for f in fuzz_fused fuzz_kron; do
  FUZZ_SECONDS=100 timeout 200 python tools/$f.py > gpurun_out/fuzz6/$f.log 2>&1; echo "$f rc=$? : $(tail -1 gpurun_out/fuzz6/$f.log | cut -c1-200)"
done
