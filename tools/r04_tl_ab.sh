for L in tl0 tl2; do
  export VLT5_LIB=$PWD/vqacl_amd/libvlt5_$L.so
  for a in "4480 3072 768 0 0 224 256" "3072 768 4480 1 1 256 256" "4480 768 768 0 1 128 64" "4480 768 3072 0 0 64 128"; do
    echo "== $L $a"; python tools/gemm_timeline.py $a 2>&1 | grep -E "per k-step|first k-tile|graph replay|prologue"
  done
done
