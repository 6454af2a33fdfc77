"""Kernel metadata (VGPRs, SGPRs, scratch bytes per lane, LDS) of every gfx950 kernel in a built library or object -- no GPU needed:
    python tools/kernel_meta.py lib.so [regex]
Splits the file's clang offload bundles by hand (roc-obj-ls needs a Perl module this image lacks) and reads the code objects' notes."""
import re, struct, subprocess, sys, tempfile, os

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", data, pos + 24)
        off = pos + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and size:
                yield data[pos + o:pos + o + size]
        pos += 24


def kernels(path):
    out = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "name" and "kernel" not in cur.get("_stage", ""):
                pass
            if k in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count", "max_flat_workgroup_size"):
                cur[k] = int(v)
            if k == "symbol":
                cur["symbol"] = v.strip("'")
            if k == "wavefront_size":
                if "symbol" in cur:
                    out.append(cur)
                cur = {}
    return out


if __name__ == "__main__":
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    for k in kernels(sys.argv[1]):
        name = subprocess.run(["c++filt", k["symbol"].replace(".kd", "")], capture_output=True, text=True).stdout.strip()
        if pat.search(name):
            print(f"{name[:100]:100s} vgpr {k.get('vgpr_count', -1):3d} sgpr {k.get('sgpr_count', -1):3d} scratch {k.get('private_segment_fixed_size', -1):5d} "
                  f"lds {k.get('group_segment_fixed_size', -1):6d}")
