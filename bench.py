#!/usr/bin/env python3
"""bench.py -- decompressed GiB/s (and % of HBM roofline) of the batched DEFLATE
extract path on MI355X: BASELINE.json's metric on its configs[1]
(10 000 synthetic enwik-style 64 KiB entries, zlib level 6, one archive image).

One "step" = one pass of the hot path over the whole batch:
ozg_decode_batch_device_ex() on an archive image, descriptor table and output
arena that are already resident in HBM (the PCIe-inclusive rate is reported
separately in DESIGN.md; it is never `value`).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: entries are independent, so there is no data-path collective.
  --scaling weak (default): each rank decodes its own 10 000-entry shard of one
      N x 10 000-entry job; RCCL carries the barrier and the max/sum reductions.
  --scaling strong: ONE 10 000-entry archive; rank 0 builds the descriptor table,
      broadcasts it over RCCL (the "work queue"), every rank takes its contiguous,
      byte-balanced run of entries (otezip_amd/dist.py) and uploads only that stretch
      of the archive; counters are all-reduced.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_shard(rank, n, size, method, cache_dir, wait_for_other=False):
    """The rank's archive image + entry table, cached on local disk between runs.
    wait_for_other: another process of this job generates this shard: wait for its files instead of
    generating the same bytes again (N ranks x 16 generator processes at once is what this avoids)."""
    import numpy as np
    from otezip_amd import corpus
    os.makedirs(cache_dir, exist_ok=True)
    tag = "m%d_n%d_s%d_r%d_v2" % (method, n, size, rank)
    p_arc, p_ent = os.path.join(cache_dir, tag + ".zip"), os.path.join(cache_dir, tag + ".npy")
    if wait_for_other:
        t0 = time.time()
        while not (os.path.exists(p_arc) and os.path.exists(p_ent)) and time.time() - t0 < 900:
            time.sleep(0.5)
    if os.path.exists(p_arc) and os.path.exists(p_ent):
        with open(p_arc, "rb") as f:
            arc = f.read()
        ent = np.load(p_ent)
        return arc, ent
    t0 = time.time()
    arc, entries = corpus.build_archive(n, size, methods=(method,), first=rank * n)
    ent = np.array([(e["method"], e["data_off"], e["comp_size"], e["uncomp_size"], e["crc32"]) for e in entries],
                   dtype=np.int64)
    log("[bench] rank %d generated %d entries in %.1fs" % (rank, n, time.time() - t0))
    tmp = p_arc + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(arc)
    os.replace(tmp, p_arc)
    np.save(p_ent + ".tmp%d.npy" % os.getpid(), ent)
    os.replace(p_ent + ".tmp%d.npy" % os.getpid(), p_ent)
    return arc, ent


def kernel_build_id():
    """Hash of the kernel sources: a PMC summary is only reported for the build it was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "otezip_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(entries, entry_bytes, method):
    """HBM bytes per launch of the DEFLATE kernels from the PMC counters.  Counters cannot be read
    from inside this process: they come from the two separate rocprofv3 --pmc passes over this very
    command that tools/pmc_traffic.py reduced to profiles/<round>/pmc_traffic.json (FETCH_SIZE x2
    + WRITE_SIZE, in KiB, as the microarchitecture guide prescribes).  None when no committed
    summary matches the workload AND the kernel sources it was measured on."""
    import glob
    best = None
    build = kernel_build_id()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
        try:
            with open(f) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        w = j.get("workload", {})
        if (w.get("entries"), w.get("entry_bytes"), w.get("method")) == (entries, entry_bytes, method) and \
                j.get("kernel_build_id") == build:
            best = int(j["traffic_bytes_per_launch"])
    return best


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(arc, ent, sample_entries):
    """Reference (oracle/_ref/ref_bench, built from /root/reference in the build container)
    timed on this box's host cores over the first `sample_entries` entries of the same
    archive; N workers with independent zip_t handles, entries i mod N.
    Falls back to the oracle restatement ("port", 1 core) when the reference build did not travel."""
    from otezip_amd import corpus  # noqa: F401
    cores = min(os.cpu_count() or 1, 16)
    ref_bench = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    n = min(sample_entries, len(ent))
    if os.path.exists(ref_bench):
        with tempfile.NamedTemporaryFile(suffix=".zip", delete=False, dir="/tmp") as f:
            f.write(arc)
            path = f.name
        try:
            t0 = time.time()
            procs = [subprocess.Popen([ref_bench, path, str(i), str(cores), str(n)], stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL) for i in range(cores)]
            outs = [p.communicate(timeout=600)[0].decode().split() for p in procs]
            wall = time.time() - t0
        finally:
            os.unlink(path)
        nbytes = sum(int(o[1]) for o in outs)
        nulls = sum(int(o[3]) for o in outs)
        worker_max = max(float(o[2]) for o in outs)
        return {"value": nbytes / worker_max / 2**30, "unit": "GiB/s", "cores": cores, "kind": "reference", "cpu": cpu_model(),
                "sample": "first %d of the same 64 KiB DEFLATE entries, zip_fopen_index+zip_fclose, %d processes "
                          "x independent zip_open, wall %.1fs, reference NULL entries %d" % (n, cores, wall, nulls)}
    from oracle import pyoracle as O
    n = min(n, 200)
    t0 = time.time()
    nbytes = 0
    for m, off, cs, us, crc in ent[:n]:
        r = O.extract_entry(int(m), arc[off:off + cs], int(us), int(crc))
        nbytes += len(r["out"])
    dt = time.time() - t0
    return {"value": nbytes / dt / 2**30, "unit": "GiB/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": "first %d entries through the C restatement oracle/otz_oracle.c (reference build absent)" % n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--entries", type=int, default=10000, help="entries per GPU")
    ap.add_argument("--entry-size", type=int, default=65536)
    ap.add_argument("--method", type=int, default=8)
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache-dir", default=os.environ.get("OTEZIP_BENCH_CACHE", "/tmp/otezip_amd_cache"))
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real runs; gloo to rehearse N>1 on one GPU")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --entries per GPU; strong: --entries in total, split over the ranks")
    args = ap.parse_args()

    from otezip_amd import dist as D
    rank, local_rank, world = D.env_world()
    if world != args.gpus:
        log("[bench] WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))

    import numpy as np
    import torch
    import otezip_amd as oz
    from otezip_amd import _native as N

    # synthetic archive first: the generator forks workers, which must happen before HIP is up
    strong = args.scaling == "strong" and world > 1
    # every rank decodes a 10 000-entry shard of its own (weak scaling); the shards are the same synthetic
    # archive -- rank 0 generates it once, the others wait for its cache files -- so that N ranks do not
    # start N x 16 generator processes at the same moment
    arc, ent = load_shard(0, args.entries, args.entry_size, args.method, args.cache_dir, wait_for_other=local_rank != 0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: otezip-amd has no CPU path")
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    D.init(args.dist_backend, dev)
    red_dev = dev if args.dist_backend == "nccl" else "cpu"  # where the scalar reductions live
    oz.init(dev_index)
    L = oz.lib()

    base_off = 0
    if strong:
        # the work queue: rank 0's descriptor table, broadcast; every rank takes its byte-balanced run
        ent = D.broadcast_table(ent if rank == 0 else None, src=0, device=red_dev)
        lo, hi = D.partition_by_bytes(ent[:, 2] + ent[:, 3], world)[rank]
        ent = ent[lo:hi]
        if len(ent):
            base_off = int(ent[0, 1])
            arc = arc[base_off:int(ent[-1, 1] + ent[-1, 2]) + 64]  # only this rank's stretch of the archive goes to its GPU
    n = len(ent)
    items = [(int(m), int(off) - base_off, int(cs), int(us), int(crc)) for m, off, cs, us, crc in ent]
    ents, dst_len = oz.make_entries(items)
    comp_bytes = int(ent[:, 2].sum())
    out_bytes = int(ent[:, 3].sum())

    d_src = torch.frombuffer(bytearray(arc), dtype=torch.uint8).to(dev)
    d_dst = torch.empty(dst_len + 64, dtype=torch.uint8, device=dev)
    d_ent = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    flags = oz.BATCH_ONLY_DEFLATE if args.method == 8 else oz.BATCH_NO_DEFLATE

    def step():
        N.check(L.ozg_decode_batch_device_ex(d_src.data_ptr(), d_src.numel(), d_dst.data_ptr(), d_dst.numel(),
                                             d_ent.data_ptr(), d_res.data_ptr(), n, stream.cuda_stream, flags), "decode")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = D.reduce_max(time.perf_counter() - t0, red_dev)
    kernel_ms = [a.elapsed_time(b) for a, b in evs]

    # every entry must have decoded and must match the directory CRC
    res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=np.int32).reshape(n, 4)
    bad = int((res[:, 0] != oz.ST_OK).sum())
    crc_bad = int(((res[:, 3].astype(np.uint32) & oz.F_CRC_MISMATCH) != 0).sum())
    f2 = int(((res[:, 3].astype(np.uint32) & oz.F_REF_FALSE_REJECT) != 0).sum())
    # and a sample is compared byte for byte with an independent decoder (host zlib)
    host = d_dst.cpu().numpy()
    mism = 0
    for i in range(0, n, max(1, n // 64)):
        m, off, cs, us, crc = items[i]
        want = zlib.decompress(arc[off:off + cs], -15) if m == 8 else None
        if want is not None and host[ents[i].dst_off:ents[i].dst_off + us].tobytes() != want:
            mism += 1
    if bad or crc_bad or mism:
        raise SystemExit("[bench] INVALID RUN: %d entries failed, %d CRC mismatches, %d byte mismatches" %
                         (bad, crc_bad, mism))

    total_out = D.reduce_sum(out_bytes, red_dev)  # whole job: every rank's shard
    if rank == 0:
        value = total_out * args.steps / elapsed / 2**30
        kavg = sum(kernel_ms) / len(kernel_ms) * 1e-3
        algo = comp_bytes + out_bytes
        line = {
            "metric": "decompressed GiB/s, batched DEFLATE decode of an HBM-resident batch (ozg_decode_batch_device_ex: the kernel span "
                      "behind zip_fopen_index), 64 KiB entries",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x %d B enwik-style entries %s, zlib-6 raw DEFLATE, ratio %.2f, "
                                   "archive image resident in HBM" % (args.entries, args.entry_size,
                                                                       "in ONE archive split over the GPUs" if strong else "per GPU (the same synthetic archive on every GPU)",
                                                                       out_bytes / max(comp_bytes, 1)),
                       "entries_per_gpu": n, "entry_bytes": args.entry_size, "method": args.method,
                       "parallelism": ("one descriptor table broadcast over RCCL, entries split %d-way by bytes" if strong else
                                       "entries sharded %d-way") % world + ", no data-path collective",
                       "ref_false_rejects": f2},
            "roofline": {"bound": "hbm", "achieved": round(algo / kavg / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(algo / kavg / HBM_PEAK, 5),
                         "traffic": pmc_traffic(n, args.entry_size, args.method),
                         "kernel": (N.lib().ozg_kernel_name_index().decode() + "+" + N.lib().ozg_kernel_name_inflate().decode() + "+otz_out_crc_kernel")
                         if args.method == 8 else N.lib().ozg_kernel_name_copy().decode(),
                         "kernel_build_id": kernel_build_id(),
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(kavg * 1e3, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(arc, ent, args.cpu_sample)
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "GiB/s", "cores": 0, "kind": "reference",
                                        "sample": "failed: %r" % (ex,)}
        print(json.dumps(line), flush=True)
    D.finish()
    oz.shutdown()


if __name__ == "__main__":
    main()
