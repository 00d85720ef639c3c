cp genima_amd/libgenima_hip.so /tmp/lib_orig.so
for v in u8 u16 u24 u8 u16 u24; do
cp genima_amd/libvariants/lib_$v.so genima_amd/libgenima_hip.so
echo "== $v"; python tools/probes/gn_bench.py 2>/dev/null | grep groupnorm | head -8
done
cp /tmp/lib_orig.so genima_amd/libgenima_hip.so
