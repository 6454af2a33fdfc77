"""Scratch (spill) stores / loads of one function of the chain kernel's translation unit, by source line (no GPU needed):
    python tools/isa_spill_lines.py <function-name-substring> [file.hip]"""
import collections, os, re, subprocess, sys
pat = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "mvmc_chain.hip"
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multiview_motion_capture_amd", "csrc")
out = "/tmp/isa_g_%s.s" % os.path.basename(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-gline-tables-only", *os.environ.get("EXTRA", "").split(),
                "-o", out, src], cwd=d, check=True, stderr=subprocess.DEVNULL)
files, fn, loc = {}, None, None
agg = collections.defaultdict(lambda: [0, 0])
for line in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"^(_Z\w+):", line)
    if m:
        fn = m.group(1)
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if fn and pat in fn:
        if "scratch_store" in line: agg[(fn[:60], loc)][0] += 1
        if "scratch_load" in line: agg[(fn[:60], loc)][1] += 1
for (f, loc), (st, ld) in sorted(agg.items(), key=lambda x: -sum(x[1]))[:40]:
    print(f, loc, "st", st, "ld", ld)
