#!/usr/bin/env python3
"""tools/kernel_regs.py [pattern ...] -- registers, spills and LDS of the built kernels (from the code object's notes), no GPU needed.
   python tools/kernel_regs.py k_bwd_mfma k_fwd_cell"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "clonealign_amd", "csrc", "build", "clonealign_hip.o")
pats = [a for a in sys.argv[1:] if not a.endswith(".o")] or [""]
for a in sys.argv[1:]:
    if a.endswith(".o"):
        obj = a
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"])
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("symbol")
    name = name[:-3] if name.endswith(".kd") else name
    try:
        name = subprocess.run([f"{LLVM}/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if any(p in name for p in pats):
        rows.append((name.split("(")[0], g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print(f"{'kernel':90s} vgpr agpr spill sgpr   lds scratch")
for r in sorted(rows):
    print(f"{r[0][:90]:90s} {r[1]:>4s} {r[2]:>4s} {r[3]:>5s} {r[4]:>4s} {r[5]:>5s} {r[6]:>7s}")
