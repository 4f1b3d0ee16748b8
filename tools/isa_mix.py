"""Instruction mix of one kernel from the gfx950 assembly: counts per class for the whole kernel and per basic block (the chunk loop
of a streaming kernel is its longest block), so that every VALU instruction of the loop can be attributed.
    python tools/isa_mix.py render.hip 'rendering_fwd_kernel<2>' [--dump]       (--dump prints the longest block's instructions)"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(ROOT, "nerfacc_amd", "csrc", sys.argv[1])
want = sys.argv[2]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-I" + os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-S", *os.environ.get("NFA_EXTRA", "").split(), "-o", out, src], stderr=subprocess.DEVNULL)
    txt = open(out).read()
def cls(i):
    if i.startswith("v_"):
        if "dpp" in i: return "valu_dpp"
        return "valu"
    if i.startswith("s_waitcnt"): return "waitcnt"
    if i.startswith("s_"): return "salu"
    if i.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if i.startswith("ds_"): return "lds"
    return "other"
for m in re.finditer(r"\n(_Z\S+):\s*; @\S+\n(.*?)\n\.Lfunc_end", txt, re.S):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    if want not in name:
        continue
    blocks, cur, label = [], [], "entry"
    for line in m.group(2).split("\n"):
        if re.match(r"^\.LBB\S+:", line):
            blocks.append((label, cur)); cur = []; label = line.split(":")[0]
        elif line.startswith("\t") and not line.strip().startswith((".", ";")):
            cur.append(line.strip())
    blocks.append((label, cur))
    tot = collections.Counter(cls(i.split()[0]) + ("" if " dpp" not in i and "row_" not in i and "wave_sh" not in i else "_dpp") if False else
                              (cls(i.split()[0]) if not (i.startswith("v_") and ("row_" in i or "wave_sh" in i or "quad_perm" in i or "row_bcast" in i)) else "valu_dpp")
                              for _, b in blocks for i in b)
    print(name.split("(")[0], "instructions", sum(tot.values()), dict(tot))
    big = sorted(blocks, key=lambda b: -len(b[1]))[:4]
    for lab, b in big:
        c = collections.Counter((cls(i.split()[0]) if not (i.startswith("v_") and ("row_" in i or "wave_sh" in i or "row_bcast" in i)) else "valu_dpp") for i in b)
        print("  block", lab, len(b), dict(c))
    if "--dump" in sys.argv:
        for lab, b in big:
            ops = collections.Counter(i.split()[0] for i in b)
            print("  opcodes of", lab, ":", ", ".join(f"{k} {v}" for k, v in ops.most_common(40)))
    if "--text" in sys.argv:
        print("\n".join(big[0][1]))
