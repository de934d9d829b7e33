"""The native writer alone on a 5 M-record callset: write-back seconds under UGVC_VCF_DEFLATE_THREADS / UGVC_VCF_LEVEL / batch size
(CPU only; run it on the GPU box for its 256 host threads).  Usage: python tools/bench_writer.py [n_variants]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from variantcalling_amd import schema as S, synth  # noqa: E402
from variantcalling_amd.io import vcf as pyvcf, vcf_native as nv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
cs = synth.make_callset(n, genome_len=400_000_000, n_contigs=4, seed=9)
src = f"/tmp/bench_writer_{n}.vcf.gz"
if not os.path.exists(src):
    pyvcf.write_vcf_from_table(src, cs.variants, cs.ref.names)
v = nv.read_vcf(src, cs.ref.names)
m = v.table.n
res = S.FilterResult(np.linspace(0, 1, m, dtype=np.float32), (np.arange(m) % 3 == 0).astype(np.uint8), np.zeros(m, np.uint8))
out = "/tmp/bench_writer_out.vcf.gz"
print(f"{m} records, {os.path.getsize(src) / 1e6:.1f} MB BGZF in, {os.cpu_count()} host threads, deflate back end {nv.set_deflate('auto')}")
for thr, lvl, batch in (("", "", ""), ("64", "", ""), ("128", "", ""), ("192", "", ""), ("256", "", ""), ("", "5", ""), ("", "4", ""), ("", "3", ""), ("", "1", ""),
                        ("", "", "262144"), ("", "", "1048576"), ("", "", "")):
    for k, val in (("UGVC_VCF_DEFLATE_THREADS", thr), ("UGVC_VCF_LEVEL", lvl), ("UGVC_VCF_WRITE_BATCH", batch)):
        if val:
            os.environ[k] = val
        else:
            os.environ.pop(k, None)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        nv.write_filtered_vcf(out, v, res)
        ts.append(time.perf_counter() - t0)
    print(f"deflate threads {thr or 'default':>7s}  level {lvl or '6':>2s}  batch {batch or '524288':>8s}: write-back {min(ts):.3f} s (median {sorted(ts)[1]:.3f}), "
          f"{os.path.getsize(out) / 1e6:.1f} MB out")
