"""Scratch (spill) traffic INSIDE LOOPS of one function of the chain kernel's translation unit (no GPU needed):
    EXTRA="-DMVMC_SMALL_WPS=4" python tools/isa_loop_spills.py <function-name-substring> [file.hip]
A loop = a backward branch to a label; for every loop (innermost first) the instruction count, the scratch stores / loads, LDS and VALU
instruction counts between the label and the branch, and the source lines (from -gline-tables-only) the scratch traffic belongs to."""
import collections, os, re, subprocess, sys
pat = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "mvmc_chain.hip"
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multiview_motion_capture_amd", "csrc")
out = "/tmp/isa_l_%s.s" % os.path.basename(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-gline-tables-only",
                *os.environ.get("EXTRA", "").split(), "-o", out, src], cwd=d, check=True, stderr=subprocess.DEVNULL)
files, fn, loc = {}, None, None
body = []   # (kind, text, loc) of the function's lines
for line in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"^(_Z\w+):", line)
    if m:
        fn = m.group(1)
        continue
    if not (fn and pat in fn):
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    m = re.match(r"^(\.LBB\w+):", line)
    if m:
        body.append(("label", m.group(1), loc))
        continue
    t = line.strip()
    if t and not t.startswith((".", ";", "//")):
        body.append(("inst", t, loc))
labels = {t: i for i, (k, t, _) in enumerate(body) if k == "label"}
loops = []
for i, (k, t, _) in enumerate(body):
    if k == "inst" and (t.startswith("s_cbranch") or t.startswith("s_branch")):
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] < i:
            loops.append((labels[tgt], i))
loops.sort(key=lambda x: x[1] - x[0])
print("function lines", len(body), "loops", len(loops))
for lo, hi in loops:
    insts = [b for b in body[lo:hi + 1] if b[0] == "inst"]
    st = [b for b in insts if "scratch_store" in b[1]]
    ld = [b for b in insts if "scratch_load" in b[1]]
    lds = sum(1 for b in insts if b[1].startswith("ds_"))
    valu = sum(1 for b in insts if b[1].startswith("v_"))
    where = collections.Counter([b[2] for b in st + ld])
    src_lo = min((b[2][1] for b in insts if b[2] and b[2][0] == insts[0][2][0]), default=0) if insts and insts[0][2] else 0
    print(f"loop @{lo}-{hi} ({insts[0][2] if insts else None}): {len(insts)} insts, valu {valu}, ds {lds}, scratch st {len(st)} ld {len(ld)}",
          dict(where.most_common(6)) if where else "")
