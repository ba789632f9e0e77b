import os, subprocess, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from otezip_amd import corpus
import otezip_amd as oz
arc, ents = corpus.build_archive(10000, 65536)
open('/tmp/big.zip','wb').write(arc)
libdir = os.path.dirname(oz.lib_path())
subprocess.check_call(["gcc", "-O1", "-std=c99", "-I", "include", "tests/c/zip_roundtrip.c", "-o", "/tmp/zr", "-L", libdir, "-lotezip_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
env = dict(os.environ, OTEZIP_TRACE="1")
for k in range(1):
    p = subprocess.run(["/tmp/zr", "bench-extract", "/tmp/big.zip", "0", "3"], capture_output=True, text=True, env=env)
    print(p.stdout.strip())
lines = p.stderr.strip().split("\n")
t0 = None
for l in lines:
    if l.startswith("[otezip-amd"):
        t = float(l.split()[1].rstrip("]"))
        t0 = t0 or t
        print("%8.1f ms  %s" % ((t - t0) * 1e3, l.split("] ", 1)[1]))
