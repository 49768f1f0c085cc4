#!/usr/bin/env python3
"""One-off validation at production plane size: five z-slab ranks of 512 x 512 x 64 (the per-rank geometry of BASELINE config 4)
on ONE GPU through the stream-ordered RCCL stand-in, against the single-GPU run of the same 512 x 512 x 320 grid (the box allows six processes on the card): the
per-slab density dumps of every frame must stitch to the single-GPU dump byte for byte."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from build_fake_rccl import build
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/six"
one, six = os.path.join(out, "one"), os.path.join(out, "six")
common = ["--grid", "512", "512", "320", "--steps", "4", "--warmup", "0", "--jacobi-iters", "60", "--no-cpu-baseline", "--no-extra"]
env = dict(os.environ, OMP_NUM_THREADS="2")
for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    env.pop(k, None)
def run(args, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "falling back" not in r.stderr, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
a = run([*common, "--dump", one], env)
b = run(["--gpus", "5", *common, "--dump", six], dict(env, BQ_RCCL_LIBRARY=build("async")))
from gpufluidsimulation_amd.solver import read_density_dump
print("one GPU:", a["ms_per_step"], "ms/step; five ranks on one GPU:", b["ms_per_step"], "ms/step, comm_size", b["config"]["comm_size"], b["config"]["grid_per_gpu"])
ok = True
for f in sorted(os.listdir(one)):
    _, rec = read_density_dump(os.path.join(one, f))
    parts = sorted(p for p in os.listdir(six) if p.startswith(f[:-4] + ".k"))
    st = np.concatenate([read_density_dump(os.path.join(six, p))[1] for p in parts])
    same = len(parts) == 5 and st.tobytes() == rec.tobytes()
    print(f, len(rec), "voxels,", len(parts), "parts,", "identical" if same else "DIFFERENT")
    ok = ok and same and len(rec) > 1000
sys.exit(0 if ok else 1)
