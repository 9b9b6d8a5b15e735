for v in "" rabl1 rabl3 rabl4 rabl5; do
  if [ -z "$v" ]; then L=""; else L="SUSTAINGYM_AMD_LIB=$PWD/sustaingym_amd/variants/lib_$v.so"; fi
  echo "${v:-regular} $(env $L python tools/rollout_greedy.py caltech 2>/dev/null | head -1 | cut -c1-120)"
done
