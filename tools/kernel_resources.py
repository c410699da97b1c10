#!/usr/bin/env python3
"""tools/kernel_resources.py [lib.so|file.o ...] [--filter substr]  --  registers, spills, scratch and LDS of every gfx950 kernel.

Reads the code object's metadata note (what `llvm-readelf --notes` prints) out of the offload bundle of a host library or object and
prints one line per kernel: VGPRs, AGPRs, spilled VGPRs / SGPRs, private segment (scratch) bytes, static LDS bytes, waves per SIMD the
register allocation allows (512 / ceil8(vgpr + agpr)).  Exit status 1 with --check-no-scratch when a kernel matching the filter has scratch.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    """gfx950 code objects inside `path` (an offload bundle in a host ELF, or a device ELF itself)."""
    out = tempfile.mkdtemp(prefix="kres_")
    dev = os.path.join(out, "dev.co")
    for kind in ("hipv4-amdgcn-amd-amdhsa--gfx950", "hip-amdgcn-amd-amdhsa--gfx950"):
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={kind}", f"--input={path}", f"--output={dev}", "--unbundle"],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(dev) and os.path.getsize(dev) > 0:
            return [dev]
    # a shared library keeps the bundle in section .hip_fatbin
    fat = os.path.join(out, "fat.bin")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True, text=True)
    if r.returncode == 0 and os.path.exists(fat) and os.path.getsize(fat) > 0:
        data = open(fat, "rb").read()
        res = []
        # the fat binary may hold several bundles back to back (one per translation unit)
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), data)]
        for k, s in enumerate(starts):
            blob = data[s: starts[k + 1] if k + 1 < len(starts) else len(data)]
            n = int.from_bytes(blob[24:32], "little")
            p = 32
            for _ in range(n):
                off = int.from_bytes(blob[p:p + 8], "little")
                size = int.from_bytes(blob[p + 8:p + 16], "little")
                tl = int.from_bytes(blob[p + 16:p + 24], "little")
                triple = blob[p + 24:p + 24 + tl].decode()
                p += 24 + tl
                if "gfx950" in triple and size:
                    f = os.path.join(out, f"dev{k}.co")
                    open(f, "wb").write(blob[off:off + size])
                    res.append(f)
        return res
    return [path]


def kernels(co):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    res = []
    for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        def g(key, default="0"):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else default
        res.append(dict(name=g("name", "?"), vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                        vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")),
                        scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size")),
                        wg=int(g("max_flat_workgroup_size"))))
    return res


def demangle(n):
    r = subprocess.run(["c++filt", n], capture_output=True, text=True)
    s = r.stdout.strip() or n
    return re.sub(r"\(.*$", "", s)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = sys.argv[sys.argv.index("--filter") + 1]
        args.remove(flt)
    check = "--check-no-scratch" in sys.argv
    paths = args or ["basicrenderer_amd/lib/libbrmi.so"]
    bad = 0
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>7s} {'waves':>5s}")
    for p in paths:
        for co in code_objects(p):
            for k in sorted(kernels(co), key=lambda k: k["name"]):
                name = demangle(k["name"])
                if flt and flt not in name:
                    continue
                alloc = max(8, -(-(k["vgpr"] + k["agpr"]) // 8) * 8)
                waves = min(8, 512 // alloc)
                print(f"{name[:70]:70s} {k['vgpr']:5d} {k['agpr']:5d} {k['sgpr']:5d} {k['vspill']:6d} {k['sspill']:6d} {k['scratch']:7d} {k['lds']:7d} {waves:5d}")
                if k["scratch"]:
                    bad += 1
    if check and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
