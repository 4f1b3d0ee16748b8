"""gpurun_out/r06_* (tools/experiments/r06_final.sh) -> the measured parts of profiles/r06_*: the bench line, the counter jsons, the kernel
summary under its header, the tables of r06_streaming.md / r06_microbench.md / r06_scene_sweep.md.  The prose of those files is written
by hand around markers: everything between `<!-- table:NAME -->` and `<!-- /table:NAME -->` is replaced.   python tools/publish_r06.py"""
import json, os, re, shutil
G, P = "gpurun_out/r06_", "profiles/r06_"
def put(path, name, text):
    s = open(path).read()
    a, b = f"<!-- table:{name} -->", f"<!-- /table:{name} -->"
    assert a in s and b in s, (path, name)
    s = s[:s.index(a) + len(a)] + "\n" + text.rstrip() + "\n" + s[s.index(b):]
    open(path, "w").write(s)
line = open(G + "bench_line.json").read().strip().splitlines()[-1]
d = json.loads(line)
open(P + "bench_line.json", "w").write(line + "\n")
shutil.copy(G + "pmc/pmc_traverse.json", P + "pmc_traverse.json")
shutil.copy(G + "pmc_1m/pmc_traverse.json", P + "pmc_traverse_1m.json")
ga, r = d["gpu_activity"], d["roofline"]
put(P + "bench_kernels.md", "line", f"**{d['value'] / 1e6:.2f} M rays/s, {d['samples_per_sec'] / 1e6:.0f} M samples/s, {d['ms_per_step']:.3f} ms per step**; `gpu_activity`: "
    f"{ga['nfa_kernels_per_step']:.2f} nfa:: launches and {ga['nfa_us_per_step']:.1f} us of nfa:: kernels per step; `roofline.avg_launch_ms` = {r['avg_launch_ms'] * 1e3:.1f} us "
    f"(HIP events around the single launch inside the timed steps, {r['launches']} launches, {r.get('emit_launches', 0)} separate emit launches); candidates per launch "
    f"{d['config']['candidate_samples_per_iter']:.0f} >= rendered samples per step {d['config']['samples_per_iter_per_gpu']:.0f}; `roofline.frac` {r['frac']:.4f}, "
    f"`roofline.path.frac` {r['path']['frac']:.4f}; `path_only_loop` {d['path_only_loop']['ms_per_step']:.3f} ms per step.")
put(P + "bench_kernels.md", "kernels", open(G + "bench_kernels_table.md").read())
put(P + "streaming.md", "stream24", open(G + "stream24.md").read())
put(P + "streaming.md", "stream18", open(G + "stream18.md").read())
u = open(G + "units.txt").read()
put(P + "streaming.md", "units", u[u.index("| kernel | us |"):] if "| kernel | us |" in u else u)
put(P + "microbench.md", "microbench", open(G + "microbench.md").read())
put(P + "microbench.md", "frames", open(G + "frame_bench.md").read())
put(P + "microbench.md", "multilevel", "```\n" + open(G + "multilevel.txt").read().rstrip() + "\n```")
s = open(G + "scene_sweep.md").read()
put(P + "scene_sweep.md", "scenes", s)
print("published r06:", d["value"], ga["nfa_us_per_step"], ga["nfa_kernels_per_step"])
