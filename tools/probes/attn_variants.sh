cp genima_amd/libgenima_hip.so /tmp/lib_orig.so
for v in bwd_f1_w1 bwd_f1_w2 bwd_f0_w2 bwd_f1_w1 bwd_f1_w2 bwd_f0_w2; do
cp genima_amd/libvariants/lib_$v.so genima_amd/libgenima_hip.so
echo "== $v"; python tools/probes/attn_bwd_bench.py 2>/dev/null | head -3
done
for v in fwd_f1_w0 fwd_f1_w3 fwd_f0_w3 fwd_f1_w0 fwd_f1_w3 fwd_f0_w3; do
cp genima_amd/libvariants/lib_$v.so genima_amd/libgenima_hip.so
echo "== $v"; python tools/bench_attn.py 2>/dev/null | head -6
done
cp /tmp/lib_orig.so genima_amd/libgenima_hip.so
