import glob, json, os, subprocess, sys
root = "/root/repo"
for rep in range(2):
    for lib in sorted(glob.glob(os.path.join(root, "variants_tmp", "lib_v*.so"))):
        env = dict(os.environ, ECCKD_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "0",
                              "--host-sample", "0", "--fortran-sample", "0"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, "fused", d["fused_lw"]["kernels_avg_ms"], "c1 %.3f c3 %.3f" % (d["configs_1"]["ms_per_step"], d["configs_3_shard"]["ms_per_step"]), flush=True)
        except Exception as e:
            print(os.path.basename(lib), "FAILED", e, out.stderr[-400:], flush=True)
