for s in "1024,1024,2048,4096" "1024,3072,4096" "2048,2048,4096" "1024,2048,4096" "512,1536,2048,4096" "1024,1024,2048,4096,8192" "1024,2048,5120" "4096"; do
  for n in 8192 32768; do
    echo -n "$s n=$n: "; GPV_HOST_CHUNKS=$s LD_LIBRARY_PATH=gnark-plonky2-verifier_amd tools/cpp/host_path_probe tests/golden/step $n 2>&1 | grep "gpv_verify    pageable"
  done
done
