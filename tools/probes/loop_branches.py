"""Per kernel of a csrc/*.hip file: conditional branches inside the loops that hold its MFMAs (what they cost was measured on the
whole call and on the attention loop: DESIGN.md, round 3).  usage: loop_branches.py genima_amd/csrc/gemm.hip [extra hipcc flags]"""
import collections, re, subprocess, sys, tempfile, os
src = sys.argv[1]; extra = sys.argv[2:]
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I", os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out] + extra,
               check=True, stderr=subprocess.DEVNULL)
txt = open(out).read().split("\n")
i = 0
while i < len(txt):
    m = re.match(r"^(_Z\S+):", txt[i])
    if not m: i += 1; continue
    name = m.group(1); j = i
    while j < len(txt) and "s_endpgm" not in txt[j] and ".Lfunc_end" not in txt[j]: j += 1
    body = txt[i:j]
    loops = collections.defaultdict(lambda: [0, 0, 0])  # header -> [mfma, cond branches, dma]
    cur = None
    for l in body:
        h = re.search(r"^\.LBB(\d+_\d+):.*(?:Loop Header|in Loop: Header=BB(\d+_\d+))", l)
        if re.match(r"^\.LBB", l):
            mm = re.search(r"in Loop: Header=BB(\d+_\d+)", l)
            hh = re.match(r"^\.LBB(\d+_\d+):.*Loop Header", l)
            cur = hh.group(1) if hh else (mm.group(1) if mm else None)
        elif re.match(r"^; %bb", l):
            mm = re.search(r"in Loop: Header=BB(\d+_\d+)", l)
            cur = mm.group(1) if mm else None
        if cur is None: continue
        if "v_mfma" in l: loops[cur][0] += 1
        if "s_cbranch" in l: loops[cur][1] += 1
        if "buffer_load" in l and " lds" in l: loops[cur][2] += 1
    hot = {h: v for h, v in loops.items() if v[0] > 0}
    if hot:
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        print(short.ljust(72), "  ".join(f"[mfma {v[0]:3d} cbranch {v[1]:2d} dma {v[2]:2d}]" for v in hot.values()))
    i = j + 1
