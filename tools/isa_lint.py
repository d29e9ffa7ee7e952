"""ISA lint of the hand-written sweeps (no GPU needed; hipcc cross-compiles): python tools/isa_lint.py [file.hip ...]

Compiles each kernel file to gfx950 assembly and reports, per kernel,
  * waterfall loops (v_readfirstlane / s_and_saveexec / s_cbranch_execnz around a buffer instruction): a buffer descriptor that the
    compiler formed in VGPRs -- every such load costs a loop per step (found in round 4 in the helper wave of the consumer / helper
    pair: 64-bit (step x size) products; fixed with running pointers);
  * `s_waitcnt vmcnt(0)` inside loops of depth >= 2 (a drain of every request in flight on every step: loads whose results meet old
    values at a join);
  * registers, scratch and occupancy.
Exit code 1 if a kernel named on the command line with --strict has a waterfall loop."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = [a for a in sys.argv[1:] if not a.startswith("--")] or sorted(glob.glob(os.path.join(ROOT, "trajoptkp_amd", "csrc", "*_mfma.hip")) +
                                                                     glob.glob(os.path.join(ROOT, "trajoptkp_amd", "csrc", "tiled_wide.hip")))
strict = "--strict" in sys.argv
bad = 0
for f in files:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",   # (the Makefile's flags)
                               "-S", "--cuda-device-only", "-o", out, f],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    cur, depth2, rows = None, False, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S+):", l)
        if m:
            cur = m.group(1); rows[cur] = dict(waterfall=0, drains=0, vgpr=0, agpr=0, scratch=0); depth2 = False
        if cur is None: continue
        if "Loop Header: Depth=" in l: depth2 = int(l.rsplit("=", 1)[1]) >= 2
        if l.startswith(".Lfunc_end"): depth2 = False
        if "s_cbranch_execnz" in l:
            back = "\n".join(lines[max(0, i - 14):i])
            # (the loop LLVM wraps a VGPR descriptor in: readfirstlane x4, v_cmp_eq_u64, s_and_saveexec, the access, s_xor exec, branch back)
            if "v_readfirstlane" in back and "s_and_saveexec" in back and "s_xor_b64 exec, exec" in back and "v_cmp_eq_u64" in back:
                rows[cur]["waterfall"] += 1
        if depth2 and "s_waitcnt vmcnt(0)" in l: rows[cur]["drains"] += 1
        for key, pat in (("vgpr", r"; NumVgprs: (\d+)"), ("agpr", r"; NumAgprs: (\d+)"), ("scratch", r"; ScratchSize: (\d+)")):
            mm = re.match(pat, l)
            if mm: rows[cur][key] = int(mm.group(1))
    print(f"{os.path.basename(f)}: {len(rows)} kernels")
    for k, r in rows.items():
        if r["waterfall"] or r["scratch"] or r["drains"]:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"   {name[:100]}: waterfall loops {r['waterfall']}, vmcnt(0) in inner loops {r['drains']}, scratch {r['scratch']} B, {r['vgpr']} VGPR + {r['agpr']} AGPR")
            bad += r["waterfall"] > 0
sys.exit(1 if strict and bad else 0)
