import sys, os, time, ctypes as C
sys.path.insert(0, "/root/repo/tools")
import extra_bench as eb
kzg = eb.load_pkg()
blob = bytes(131072)
p1 = kzg.BlstP1()
kzg.compute_challenge(blob, p1)
t0 = time.perf_counter()
for _ in range(200):
    kzg.compute_challenge(blob, p1)
dt = (time.perf_counter() - t0) / 200
print("compute_challenge: %.1f us -> %.2f GB/s" % (dt * 1e6, 131072 / dt / 1e9))
print(open("/proc/cpuinfo").read().count("sha_ni"), "cpus with sha_ni;", os.cpu_count(), "cpus")
