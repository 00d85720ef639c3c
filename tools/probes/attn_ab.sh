# the self-attention shapes of the call with two library builds, alternating (one box):  bash tools/probes/attn_ab.sh <variant.so>
V=${1:?variant library}
for i in 1 2 3; do
  echo "base:    $(python tools/bench_attn.py 2>/dev/null | grep -E 'Nq=(4096|1024) Nk=(4096|1024)' | grep -v row-major | awk '{print $(NF-3), $(NF-1)}' | tr '\n' ' ')"
  echo "variant: $(GN_LIB_PATH=$V python tools/bench_attn.py 2>/dev/null | grep -E 'Nq=(4096|1024) Nk=(4096|1024)' | grep -v row-major | awk '{print $(NF-3), $(NF-1)}' | tr '\n' ' ')"
done
