"""The profiled pass of bench.py (`gpu_activity`, `path_us_per_step`, `gpu_idle_frac`) and the DDA step count of a ray batch."""
import torch


def profile_steps(step_fn, n_steps):
    """{'busy_us_per_step', 'nfa_us_per_step', 'kernels_per_step', 'nfa_kernels_per_step', 'top'} from torch.profiler's
    device-kernel events, or None when the profiler is unavailable"""
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(n_steps):
                step_fn()
            torch.cuda.synchronize()
        ivals, nfa_us, n_nfa, per_name = [], 0.0, 0, {}
        for ev in prof.events():
            if getattr(ev, "device_type", None) is None or "cuda" not in str(ev.device_type).lower():
                continue
            dur = float(getattr(ev, "device_time", 0.0) or getattr(ev, "cuda_time", 0.0) or 0.0)
            if dur <= 0.0:
                continue
            start = float(ev.time_range.start)
            ivals.append((start, start + dur))
            short = ev.name.split("(")[0].replace("void ", "")[:70]
            c = per_name.setdefault(short, [0, 0.0])
            c[0] += 1
            c[1] += dur
            if "nfa::" in ev.name:
                nfa_us += dur
                n_nfa += 1
        if not ivals:
            return None
        ivals.sort()
        busy, (cs, ce) = 0.0, ivals[0]
        for a, b in ivals[1:]:
            if a > ce:
                busy += ce - cs
                cs, ce = a, b
            else:
                ce = max(ce, b)
        busy += ce - cs
        top = sorted(per_name.items(), key=lambda kv: -kv[1][1])[:8]
        return {"busy_us_per_step": busy / n_steps, "nfa_us_per_step": nfa_us / n_steps,
                "kernels_per_step": len(ivals) / n_steps, "nfa_kernels_per_step": n_nfa / n_steps,
                "top_kernels_us_per_step": {k: round(v[1] / n_steps, 2) for k, v in top}}
    except Exception as e:      # noqa: BLE001  (a missing profiler must not cost the bench line)
        return {"error": repr(e)[:200]}


def dda_steps(rays_o, rays_d, aabb, res, near, far):
    """voxels a ray's DDA walk visits in a one-level grid without early termination: 1 + L1 distance between the
    first and the last voxel (utils_grid.cuh:58-142: the walk ends when an index passes final_index)"""
    lo, hi = aabb[:3], aabb[3:]
    inv = 1.0 / rays_d
    t0, t1 = (lo - rays_o) * inv, (hi - rays_o) * inv
    tmin = torch.minimum(t0, t1).amax(-1).clamp_min(near)
    tmax = torch.maximum(t0, t1).amin(-1).clamp_max(far)
    ok = tmax > tmin
    cell = lambda t: (((rays_o + rays_d * t[:, None]) - lo) / (hi - lo) * res).floor().clamp(0, res - 1)
    steps = 1 + (cell(tmax - 1e-6) - cell(tmin + 1e-6)).abs().sum(-1)
    return int(steps[ok].sum().item())
