"""A/B sweep over engine env overrides; prints it/s and per-kernel us/iter (profiled warm-up).
Round 5: the RELEASE library reads no switches from the environment -- this tool needs the timing-lab build (make -C clonealign_amd/csrc lab;
CLONEALIGN_HIP_LIB=build_ab/libclonealign_hip_lab.so, which bench.py takes only with --allow-foreign-lib, passed below); for the shipped library
use bench.py --variant-off / --variant-on / --tune, or tools/stair_time.py --ab=<variants>."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(env, extra=()):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()}); e["CLONEALIGN_DEBUG_ENV"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--busy-seconds", "0", "--allow-foreign-lib", *extra],
                         env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    k = d["kernel_ms_per_iter_warmup"]
    print(env, extra, f"{d['value']:.0f} it/s", d["config"]["y_storage"], {n: round(v * 1e3) for n, v in k.items()}, flush=True)
if __name__ == "__main__":
    import ast
    extra = ()
    for arg in sys.argv[1:]:
        if arg.startswith("--"):
            extra = tuple(arg.split("="))
            continue
        run(ast.literal_eval(arg), extra)
