"""Per-kernel register / LDS / spill table of every gfx950 kernel in magicdec_amd/csrc (compile-only, no GPU needed):
hipcc -S each .hip with the Makefile's flags and read the .amdhsa metadata.  The occupancy claims in DESIGN.md
(2 workgroups of 4 waves per CU for the attention kernels, no scratch in any steady-state loop) are checked against this.

    python tools/kernel_resources.py [> profiles/r01_kernel_resources.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "magicdec_amd", "csrc")
EXACT = {"kvops", "streaming", "snapkv", "elementwise"}      # Makefile: -ffp-contract=off for the bit-exact kernels


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def main():
    rows = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        base = f[:-4]
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, base + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                   os.path.join(CSRC, f), "-o", out]
            if base in EXACT:
                cmd.insert(3, "-ffp-contract=off")
            subprocess.run(cmd, check=True, capture_output=True)
            txt = open(out).read()
        for blk in re.findall(r"- \.agpr_count.*?\.wavefront_size", txt, flags=re.S):
            g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk).group(1)
            rows.append((base, g("name"), int(g("vgpr_count")), int(g("agpr_count")), int(g("sgpr_count")),
                         int(g("vgpr_spill_count")), int(g("sgpr_spill_count")), int(g("group_segment_fixed_size")),
                         int(g("private_segment_fixed_size")), int(g("max_flat_workgroup_size"))))
    names = demangle([r[1] for r in rows])
    print(f"{'file':12} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'vspill':>6} {'sspill':>6} {'lds_static':>10} {'scratch':>7} "
          f"{'wg':>5}  kernel")
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)
        print(f"{r[0]:12} {r[2]:4d} {r[3]:4d} {r[4]:4d} {r[5]:6d} {r[6]:6d} {r[7]:10d} {r[8]:7d} {r[9]:5d}  {n}")
    clean = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n))
    print(f"\n{len(rows)} kernels; kernels with VGPR spills: {[clean(n) for r, n in zip(rows, names) if r[5]] or 'none'}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
