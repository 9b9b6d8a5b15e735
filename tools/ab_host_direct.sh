for n in 64 256 1024 4096 16384; do
  for m in 0 1073741824; do
    echo "N=$n direct_max=$m $(EVC_HOST_DIRECT_MAX_BYTES=$m python tools/numpy_path_profile.py $n 2>/dev/null | head -2 | tr '\n' ' ')"
  done
done
