"""Transposed counter table (one column per kernel, shares of SQ_WAVE_CYCLES) from the summary.csv of scripts/pmc_summarise.py.
usage: python scripts/pmc_table.py <summary.csv> > table.md"""
import csv, sys
rows = {r['kernel']: r for r in csv.DictReader(open(sys.argv[1]))}
want = ['attn_cluster_fwd_k<true, 2>', 'attn_cluster_bwd_k<true, 2>', 'gru256_cluster_fwd_k', 'gru256_cluster_bwd_k', 'gru128_seq_fwd_k<512>',
        'gru128_seq_bwd_k<512>', 'highway4_fwd_k', 'highway4_bwd_k', 'conv_gemm_tn2_group<64, 64, 32, 3>', 'conv_gemm_nt2<64, 64, 32, 3, false>',
        'conv_gemm_nn2<64, 64, 32, 3, false>', 'conv_gemm_tn3_group', 'conv_gemm_nt3<false>']
if len(sys.argv) > 2:                  # restrict to kernels whose name contains one of the given substrings
    want = [k for k in rows if any(a in k for a in sys.argv[2:])]
ks = [k for k in want if k in rows]
ctrs = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
        'SQ_ACTIVE_INST_SCA', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD',
        'SQ_INSTS_VMEM_WR', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS', 'FETCH_SIZE', 'WRITE_SIZE']
share = {'SQ_BUSY_CU_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA',
         'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAIT_INST_LDS'}
print('| counter | ' + ' | '.join(ks) + ' |')
print('|' + '---|' * (len(ks) + 1))
print('| launches per step | ' + ' | '.join('%g' % float(rows[k]['launches_per_step']) for k in ks) + ' |')
for c in ctrs:
    cells = []
    for k in ks:
        v = float(rows[k].get(c, 0) or 0)
        wc = float(rows[k].get('SQ_WAVE_CYCLES', 0) or 0)
        cells.append('%.3g' % v + (' (%.1f %% wc)' % (100 * v / wc) if c in share and wc else ''))
    print('| %s | ' % c + ' | '.join(cells) + ' |')
