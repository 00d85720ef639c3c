cp genima_amd/libgenima_hip.so /tmp/lib_orig.so
for v in pipe_w1 pipe_w3 pipe_w1 pipe_w3; do
cp genima_amd/libvariants/lib_$v.so genima_amd/libgenima_hip.so
echo "== $v default"; python tools/bench_attn.py 2>/dev/null | grep -v row-major | head -2
echo "== $v GN_ATTN_VARIANT=3"; GN_ATTN_VARIANT=3 python tools/bench_attn.py 2>/dev/null | grep -v row-major | head -2
done
cp /tmp/lib_orig.so genima_amd/libgenima_hip.so
