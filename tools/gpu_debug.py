"""Ad-hoc GPU diagnostic (not a test): prints per-entry status words for a few seeded cases."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
import otezip_amd as oz
from test_gpu_parity import run_gpu
oz.init(0)
names = sys.argv[1:] or ["deflate_valid"]
for name in names:
    items = getattr(cases, name)()
    b = run_gpu(oz, items)
    for i, it in enumerate(items):
        r = b.results[i]
        if r.status == -30:
            print(name, i, "usize", it[2], "comp", len(it[1]), "INTERNAL reason", r.flags >> 28, "lane", r.total_out, "pos", r.crc32, "aux", hex(r.flags & 0x0FFFFFFF), it[1][:8].hex())
    from collections import Counter
    print(name, Counter(b.results[i].status for i in range(len(items))))
