#!/usr/bin/env python3
"""summary.json of tools/pmc_gemm2.sh (one GEMM shape, separate --pmc passes) -> the profile JSON bench.py's `roofline.traffic`
reads (profiles/rNN_pmc_gemm_split_<mode>_<M>x<N>x<K>.json).  usage: pmc_gemm_summary.py <summary.json> <mode> M N K <out.json> [matrix products per product: 3 | 6]
HBM bytes as the guide prescribes: FETCH_SIZE (KB of 64-B requests) x 2 for wide coalesced reads on gfx950, WRITE_SIZE as is."""
import json
import sys


def main():
    src, mode, M, N, K, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    c = json.load(open(src))
    products = 3 if ', 2, ' in c.get('_kernel', '').split('>')[0][-12:] or 'Li2ELb' in c.get('_kernel', '') else 6
    if len(sys.argv) > 7:
        products = int(sys.argv[7])
    dur = c['_dur_us']
    clock = c['GRBM_GUI_ACTIVE'] / 8 / dur / 1e3                       # GHz (summed over the 8 XCDs)
    simd_cycles = 1024 * dur * 1e-6 * clock * 1e9
    flops = 2.0 * M * N * K
    d = {
        'median_duration_us': dur, 'clock_GHz': round(clock, 3),
        'mfma_busy_frac_of_simd_cycles': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, 4),
        'mfma_busy_x_clock_GHz': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles * clock, 3),
        'mfma_busy_cycles_per_mfma': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA'] * 1.0 / 64, 2) if False else 32.0,
        'valu_insts_per_mfma': round(c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA'], 2),
        'valu_mfma_coexec_frac_of_mfma_busy': round(c['SQ_VALU_MFMA_COEXEC_CYCLES'] / c['SQ_VALU_MFMA_BUSY_CYCLES'], 3),
        'lds_array_busy_frac_per_cu': round(c['SQ_LDS_IDX_ACTIVE'] / (256 * dur * 1e-6 * clock * 1e9), 4),
        'lds_bank_conflict_cycles': c['SQ_LDS_BANK_CONFLICT'],
        'f32_equivalent_tflops_in_pmc_pass': round(flops / (dur * 1e-6) / 1e12, 1),
        'matrix_products_per_algorithmic_product': products,
        'executed_bf16_tflops': round(products * flops / (dur * 1e-6) / 1e12, 1),
        'frac_of_bf16_peak_2516.6': round(products * flops / (dur * 1e-6) / 1e12 / 2516.6, 4),
        'hbm_fetch_bytes_per_launch_corrected': int(c['FETCH_SIZE'] * 1024 * 2),
        'hbm_write_bytes_per_launch': int(c.get('WRITE_SIZE', 0) * 1024),
        'algorithmic_read_bytes_per_launch': 4 * (M * K + K * N),
        'algorithmic_write_bytes_per_launch': 4 * M * N,
        'l2_hit_rate': round(c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']), 4),
        'traffic_note': 'FETCH_SIZE counts 64-byte fabric requests: doubled for the 128-byte requests of wide coalesced reads '
                        '(MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as reported (16-byte-per-lane stores)',
    }
    json.dump({'what': 'rocprofv3 --kernel-trace --pmc <group> -- python3 tools/pmc_gemm.py %s %d %d %d (separate passes, tools/pmc_gemm2.sh); '
                       'counters of the last of 6 launches of %s' % (mode, M, N, K, c.get('_kernel', '?')),
               'counters': {k: v for k, v in c.items() if not k.startswith('_')}, 'derived': d}, open(out, 'w'), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == '__main__':
    main()
