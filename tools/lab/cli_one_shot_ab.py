import subprocess, time, os, sys
def run(env):
    t = time.time()
    r = subprocess.run(["bin/regtools-amd", "junctions", "extract", "-s", "XS", "-o", "/tmp/o.bed", "/tmp/b.bam"], stderr=subprocess.PIPE, env=dict(os.environ, REGTOOLS_AMD_STATS="1", **env))
    w = time.time() - t
    line = [l for l in r.stderr.decode().splitlines() if "process:" in l]
    return w, line[-1] if line else ""
for k in range(4):
    for name, env in (("one-shot", {}), ("streams", {"REGTOOLS_AMD_ONE_SHOT": "0"})):
        w, l = run(env)
        print("%-9s wall %.4f  %s" % (name, w, l[22:]))
