"""HBM traffic per launch of the timed kernel families from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_step.sh
(set4 / set5) -> profiles/rNN_pmc_traffic.json, which bench.py quotes as `roofline.traffic`.
Units per MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request -> x 2.
usage: python scripts/pmc_traffic.py <pmc dir> [steps=3] [out.json]"""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join('profiles', 'r03_pmc_traffic.json')
FAMILIES = collections.OrderedDict([
    ('dW GEMM (conv_gemm_tn2_group)', ('conv_gemm_tn3_group', 'conv_gemm_tn2_group')), ('dX GEMM (conv_gemm_nt2)', ('conv_gemm_nt2', 'conv_gemm_nt3')),
    ('fwd GEMM (conv_gemm_nn2)', 'conv_gemm_nn2'), ('attention recurrence bwd (attn_cluster_bwd_k)', 'attn_cluster_bwd_k'),
    ('attention recurrence fwd (attn_cluster_fwd_k)', 'attn_cluster_fwd_k'),
    ('decoder GRU(256) bwd (gru256_cluster_bwd_k)', 'gru256_cluster_bwd_k'), ('decoder GRU(256) fwd (gru256_cluster_fwd_k)', 'gru256_cluster_fwd_k'),
    ('biGRU(128) bwd (gru128_seq_bwd_k)', 'gru128_seq_bwd_k'), ('biGRU(128) fwd (gru128_seq_fwd_k)', 'gru128_seq_fwd_k'),
    ('highway x4 bwd (highway4_bwd_k)', 'highway4_bwd_k'), ('highway x4 fwd (highway4_fwd_k)', 'highway4_fwd_k')])
SRC = (out + ' <- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, %d eager C2 steps incl. the first, '
       'scripts/pmc_step.sh + scripts/pmc_traffic.py); KiB x 1024, FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request)' % steps)


def sums(counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(root, 'set*', '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            for fam, key in FAMILIES.items():
                if any(k in r['Kernel_Name'] for k in ((key,) if isinstance(key, str) else key)):
                    tot[fam] += float(r['Counter_Value']); n[fam] += 1
    return tot, n


fetch, nf = sums('FETCH_SIZE')
write, nw = sums('WRITE_SIZE')
res = collections.OrderedDict()
for fam in FAMILIES:
    if not nf[fam] or not nw[fam]:
        continue
    fb, wb = fetch[fam] * 1024.0 * 2.0 / nf[fam], write[fam] * 1024.0 / nw[fam]
    res[fam] = {'bytes_per_launch': fb + wb, 'fetch_bytes_per_launch_corrected_x2': fb, 'write_bytes_per_launch': wb,
                'launches_per_step': nf[fam] / steps, 'bytes_per_step': (fb + wb) * nf[fam] / steps, 'source': SRC}
json.dump(res, open(out, 'w'), indent=1)
for k, v in res.items():
    print('%-50s %8.1f MB/launch (fetch %.1f, write %.1f) x %.1f launches/step' % (k, v['bytes_per_launch'] / 1e6,
          v['fetch_bytes_per_launch_corrected_x2'] / 1e6, v['write_bytes_per_launch'] / 1e6, v['launches_per_step']))
