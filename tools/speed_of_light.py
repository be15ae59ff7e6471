#!/usr/bin/env python
"""Speed-of-light table of the QM9 training step: for every kernel of a step budget
(tools/step_profile.py output) the floor  max(FLOPs / MFMA peak, bytes / HBM peak, launches x 1.5 us)  next to the
measured time per step.

    python tools/speed_of_light.py gpurun_out/step_budget.txt [n eg el tp L] > profiles/r04_speed_of_light.txt

Peaks (MI355X_MICROARCH.md): fp32 MFMA 157.3 TFLOP/s -- the precision the path is priced on; kernels that issue
their products on the bf16 pipe as six exact piece products ("bf16x6") have the higher ceiling 2 500 / 6 = 417 TFLOP/s,
printed beside it -- HBM 8 TB/s, and 1.5 us per dependent launch (the guide's kernel-boundary price).  FLOPs / bytes are
ALGORITHMIC (2 d^2 per row and dense layer; every operand tensor read once, every result written once, gathers counted
as the rows they deliver), for the shapes of the bench batch: n nodes, eg / el global / local edges, tp triplet + pair rows."""
import re
import sys

FP32, BF16X6, HBM, LAUNCH = 157.3e12, 2500e12 / 6, 8e12, 1.5e-6


def model(n, eg, el, tp, L, d=128):
    row = 4.0 * d                      # bytes of one fp32 row
    dense = lambda rows, layers: 2.0 * d * d * rows * layers
    own_l = 5 * n + 4 * el + 2 * tp    # rows of a local layer's own weight-gradient jobs
    own_g = 3 * n + 2 * eg
    # A node chain is `layers` DEPENDENT dense layers on a 16-row tile per workgroup (ceil(n / 16) = 143 of the 256 CUs have a
    # tile): whatever the chip's peak, a layer cannot finish before one CU has issued its 16 x 128 x 128 tile -- 0.85 us on the
    # fp32 matrix pipe.  That serial term is the chain launches' floor (entries carry it as a 6th field).
    layer_us = 2.0 * 16 * d * d / (FP32 / 256) * 1e6
    chain = lambda layers: layers * layer_us * 1e-6
    return [
        # (name substring, FLOPs per launch, bytes per launch, bf16x6?, what[, serial seconds per launch])
        ('wgrad_fused_kernel', dense(own_l + own_g, 1), row * 2 * (own_l + own_g), True, 'dW of a layer pair (16 jobs) + reductions'),
        ('wgrad_fused_wide_kernel', dense(own_g + 10 * n, 1), row * 2 * (own_g + 10 * n), True, 'dW of the last global layer'),
        ('node_tail_bwd_kernel<true, false, true>', dense(n, 7 + 1 + 3) + dense(10 * n, 1), row * n * (10 + 10 + 4 + 6) + row * 20 * n, False,
         'chain backward (7 layers) + next head backward (1 + 2..4 blocks) + 10 riding dW jobs', chain(7 + 4)),
        ('node_tail_fwd_kernel<true, false, true>', dense(n, 7 + 1 + 3) + dense(tp / 2, 2), row * n * (2 + 10 + 4) + row * tp / 2 * 4, False,
         'chain forward (7 layers) + next head (1 + 2..4 blocks) + half a triplet/pair MLP riding', chain(7 + 4)),
        ('node_tail_fwd_kernel<true, false, false>', dense(n, 7), row * n * 12, False, 'last chain forward', chain(7)),
        ('node_tail_bwd_kernel<true, false, false>', dense(n, 7), row * n * 20, False, 'first chain backward', chain(7)),
        ('local_bwd_pair_kernel', dense(tp, 2) + dense(el, 4), row * (tp * 6 + el * 11), True, 'mlp_sbf backward + local edge backward'),
        ('global_edge_agg_bwd_kernel', dense(eg, 2), row * (eg * 6 + n * 2), True, 'global edge MLP backward + target-side sum'),
        ('global_edge_agg_fwd_kernel', dense(eg, 2), row * (eg * 3 + n * 4), True, 'global edge MLP -> node segment-sum'),
        ('local_edge_fwd_kernel', dense(el, 4), row * (el * 7 + n * 4), True, 'local edge stage forward'),
        ('local_agg_bwd_kernel', 0, row * (2 * tp + 8 * el + n), False, 'rows -> edges -> nodes aggregation backward'),
        ('local_agg_fwd_kernel', 0, row * (2 * tp + 3 * el + 2 * n), False, 'rows -> edges -> nodes aggregation'),
        ('embed_multi_bwd_kernel', 2.0 * d * (16 * (eg + el) + 42 * tp) * 2, row * (eg + el + tp) + 4.0 * 42 * tp, False, 'input embeddings backward'),
        ('embed_multi_fwd_kernel', 2.0 * d * (16 * (eg + el) + 42 * tp), row * (eg + el + tp) + 4.0 * 42 * tp, False, 'input embeddings'),
        ('node_heads_bwd_kernel', dense(n, 3) * 2 * L, row * n * 8 * 2 * L, False, 'head branches of all 2L chains, backward (3 dependent layers)', chain(3)),
        ('node_heads_fwd_kernel', dense(n, 3) * 2 * L, row * n * 5 * 2 * L, False, 'head branches of all 2L chains (3 dependent layers)', chain(3)),
        ('segment_sum_split_kernel<32, 4, false, false, false, true>', 0, row * (eg + n) + 4.0 * eg, False, 'source-side sum (transposed CSR)'),
        ('segment_sum_multi_kernel', 0, row * (4 * el + 4 * n), False, 'four local segment sums'),
        ('mlp2_fwd_kernel', dense(tp, 2), row * tp * 4, True, 'first triplet/pair MLP'),
        ('adam_ema_kernel', 0, None, False, 'clip + Adam + EMA + zero_grad'),
        ('mol_graph_kernel', 0, 0, False, 'graph construction (side stream)'),
    ]


def main(path, n=2286, eg=32888, el=4316, tp=17640, L=6, params=3581100):
    rows = []
    for line in open(path):
        m = re.match(r'^(.*?)\s+(\d+\.\d)\s+(\d+\.\d)\s+(\d+\.\d+)\s*$', line.rstrip('\n'))
        if m and not line.startswith('kernel'):
            rows.append((m.group(1).strip(), float(m.group(2)), float(m.group(3)), float(m.group(4))))
    head = open(path).readline().strip()
    mdl = model(n, eg, el, tp, L)
    print('speed of light of the QM9 training step (n=%d, E_g=%d, E_l=%d, T+P=%d, L=%d, d=128)' % (n, eg, el, tp, L))
    print('source budget: ' + head)
    print('floor = max(FLOPs / 157.3 TFLOP/s fp32 MFMA, bytes / 8 TB/s, launches x 1.5 us, dependent layers x 0.85 us for the node chains); '
          '[bf16x6]: FLOP term at 417 TFLOP/s')
    print('%-58s %6s %9s %9s %9s %7s  %s' % ('kernel', 'calls', 'us/step', 'floor us', '[bf16x6]', 'x floor', 'what'))
    tm = tf = tf6 = 0.0
    for name, calls, us, avg in rows:
        hit = next((x for x in mdl if name.startswith(x[0]) or x[0] in name), None)
        serial = 0.0
        if hit is None:
            fl = by = 0.0
            six, what = False, ''
        else:
            fl, by, six, what = hit[1:5]
            serial = hit[5] if len(hit) > 5 else 0.0
            if by is None:
                by = 9 * 4.0 * params
        f32 = max(fl / FP32, by / HBM, LAUNCH, serial) * calls * 1e6
        f6 = max(fl / (BF16X6 if six else FP32), by / HBM, LAUNCH, serial) * calls * 1e6
        tm, tf, tf6 = tm + us, tf + f32, tf6 + f6
        print('%-58s %6.1f %9.1f %9.1f %9.1f %7.1f  %s' % (name[:58], calls, us, f32, f6, us / f32 if f32 else 0.0, what))
    print('%-58s %6s %9.1f %9.1f %9.1f %7.1f' % ('total (kernels listed)', '', tm, tf, tf6, tm / tf))
    print('The dependent chain alone -- %d launches on the critical path -- is %.0f us of launch boundaries.' % (
        int(sum(r[1] for r in rows)), sum(r[1] for r in rows) * 1.5))


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0], *[int(v) for v in a[1:6]])
