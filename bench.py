#!/usr/bin/env python3
"""Headline benchmark: NormalizingFlow.log_prob throughput, BASELINE.json cfg 2 (configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh child
processes of `torch.distributed.run`, before this process has touched a GPU) and relays rank 0's JSON line.

One "step" = one pass of the hot path over one synthetic batch per rank: log_prob of x[2^20, 64]
(standard normal, stored bf16, resident in HBM) through 8 alternating affine couplings (H = 64) and the
UnitNormal base density, summed to one fp64 on the device and, for N > 1, all-reduced over RCCL (the
path's only exchange).  Weak scaling: every rank owns 2^20 rows (cfg 5 = 8 x 2^20).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant (only) kernel of a step, the fused flow
kernel: algorithmic flops (98,304 per row) against the dense MFMA peak of the dtype its GEMMs
issue in (fp16, three split products per algorithmic product); `roofline_elementwise` is the standalone HBM-bound
affine coupling kernel (north_star: "achieved HBM GB/s on the element-wise path"); `configs` (N = 1 only) carries the
same measurement for BASELINE cfg 3 (8 rational-quadratic spline couplings) and cfg 4 (AffineLU + MatrixExponential +
couplings, D = 128) at 2^20 rows, timed in this process after the headline; `cpu_baseline` is the oracle (a torch-CPU
port that follows the reference op for op, incl. its double conditioner call) timed on this box's host cores on a
bounded sample: a pool of floor(usable cores / best thread count) worker PROCESSES, pinned to disjoint cores, each over its own
row block, released together (`cores` = processes x threads; `host_cores_available` = the affinity mask capped by the cgroup
CPU quota -- the GPU box shows 256 CPUs and grants 16).  The pool is started before this process touches the GPU and sits
idle during the timed region.  Fields ending in `_pmc` / `traffic` come from committed rocprofv3 --pmc passes
(profiles/*.json) and are quoted only when their build id equals the library's.  The LAST keys of the line are the first-level
scalars `value_* / ms_per_step_* / roofline_frac_*` of cfg2_exact, cfg3 and cfg4 (they must survive a 2,000-character tail).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM, LAYERS, HIDDEN = 64, 8, 64
ROWS_PER_GPU = 1 << 20
FLOPS_PER_ROW = LAYERS * 2 * (DIM // 2 * HIDDEN + HIDDEN * DIM)       # 98,304 (pruned)
BYTES_PER_ROW = DIM * 2 + 4                                            # 132 B (bf16 x in, fp32 log_prob out)
PEAK_F32_MFMA_TFLOPS = 157.3                                           # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0                                          # MI355X_MICROARCH.md, dense BF16/FP16 MFMA
PEAK_HBM_GBS = 8000.0                                                  # MI355X_MICROARCH.md, HBM3E spec
ELEMWISE_BYTES_PER_ROW = 128 + 256 + 128 + 8                           # 520 B/row/layer (bf16 x,y)
# SURVEY 8(d) per-row figures of the other single-GPU configurations
CFG3_FLOPS_PER_ROW, CFG3_BYTES_PER_ROW = 1572864, 260
CFG4_FLOPS_PER_ROW, CFG4_FLOPS_PER_ROW_COLLAPSED, CFG4_BYTES_PER_ROW = 589824, 458752, 516


def spawn_ranks(args) -> int:
    """--gpus N > 1 without a launcher: run N fresh ranks under torch.distributed.run and relay their output.
    Nothing in this process has initialised the GPU at this point (no exec of an initialised process either)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__),
           '--gpus', str(args.gpus), '--steps', str(args.steps), '--warmup', str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append('--no-cpu-baseline')
    if getattr(args, 'backend', 'nccl') != 'nccl':
        cmd += ['--backend', args.backend]
    return subprocess.run(cmd, env=env).returncode


def event_ms(fn, reps, inner=8):
    """Average device time of one launch of `fn` by HIP events on the current stream: `reps` groups of
    `inner` back-to-back launches, each group bracketed by two events (amortises the event overhead)."""
    import torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    fn()
    for a, b in evs:
        a.record()
        for _ in range(inner):
            fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) / inner for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def usable_cores():
    """Cores this process may actually keep busy: its affinity mask, capped by the cgroup's CPU quota (a container that sees 256
    cores but holds a quota of N is throttled once more than N threads spin)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = max(1, min(n, q // per))
        except Exception:
            pass
    return n


def _oracle_probe(spec, dim, probe_rows, avail, budget_s):
    """Thread count of ONE oracle process: torch's intra-op pool gets slower beyond a few dozen threads on these shapes (256
    threads measured 50x slower than 16), so walk up from 8 and stop at the first count that is not faster."""
    import torch
    from oracle import stribor_oracle as orc
    gen = torch.Generator().manual_seed(1234)
    best = None
    with torch.no_grad():
        x = torch.randn(probe_rows, dim, generator=gen)
        for th in [c for c in (8, 16, 32, 64) if c <= avail] or [avail]:
            torch.set_num_threads(th)
            orc.flow_log_prob(spec, x[:max(256, probe_rows // 16)])    # warm-up
            t0 = time.perf_counter()
            orc.flow_log_prob(spec, x)
            dt = time.perf_counter() - t0
            if best is not None and dt >= best[1]:
                break
            best = (th, dt)
            if dt > 0.5 * budget_s:
                break
    return best


def cpu_worker(argv):
    """`bench.py --cpu-worker <spec.pt> <mode> ...` (never touches the GPU).  mode `probe`: print the best thread count of one
    process; mode `run <threads> <rows> <passes> <first_core>`: pin to `threads` cores, warm up, print READY, wait for a line on
    stdin, run `passes` passes of oracle.flow_log_prob over `rows` rows, print the elapsed time."""
    import torch
    from oracle import stribor_oracle as orc
    from stribor_amd.util import flowdesc as fd
    job = torch.load(argv[0])
    spec = fd.flow_spec(job['desc'], job['state'])
    dim = job['dim']
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    if argv[1] == 'probe':
        probe_rows, budget_s = int(argv[2]), float(argv[3])
        th, dt = _oracle_probe(spec, dim, probe_rows, min(len(avail), usable_cores()), budget_s)
        print(json.dumps({'threads': th, 'dt': dt, 'rows': probe_rows}), flush=True)
        return 0
    threads, rows, passes, first = int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
    if hasattr(os, 'sched_setaffinity') and first >= 0:
        os.sched_setaffinity(0, set(avail[first:first + threads]) or set(avail))
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(1234 + max(first, 0))
    with torch.no_grad():
        x = torch.randn(rows, dim, generator=gen)
        orc.flow_log_prob(spec, x[:max(256, rows // 16)])
        print('READY', flush=True)
        sys.stdin.readline()
        t0 = time.perf_counter()
        for _ in range(passes):
            orc.flow_log_prob(spec, x)
        dt = time.perf_counter() - t0
    print(json.dumps({'dt': dt, 'rows': rows * passes}), flush=True)
    return 0


def cpu_pool(_argv):
    """`bench.py --cpu-pool`: started by main() BEFORE the bench process touches the GPU (a GPU-initialised process must not fork +
    exec on the GPU boxes); imports no torch, answers one JSON request per stdin line by running oracle workers as its own children:
    a thread-count probe in one process, then floor(available cores / threads) pinned worker processes over disjoint row blocks,
    released together; the answer carries the aggregate rows/s between the release and the last worker's end."""
    me = os.path.abspath(__file__)
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        if req.get('quit'):
            break
        try:
            avail = usable_cores()
            base = [sys.executable, me, '--cpu-worker', req['spec']]
            t_req = time.perf_counter()
            pr = json.loads(subprocess.run(base + ['probe', str(req['probe_rows']), str(req['budget_s'])], capture_output=True, text=True,
                                           timeout=120, check=True).stdout.strip().splitlines()[-1])
            print(f'[cpu pool] probe: {pr} ({time.perf_counter() - t_req:.1f} s), usable cores {avail}', file=sys.stderr, flush=True)
            th = pr['threads']
            rate1 = pr['rows'] / pr['dt']                                       # one process, best thread count
            rows = min(req['rows_per_pass'], max(256, 1 << int(rate1 * req['budget_s']).bit_length() - 1))
            passes = max(1, int(rate1 * req['budget_s'] / rows))
            workers = max(1, avail // th)
            try:                                                                 # stay far from the box's memory: ~40 fp32 [rows, dim]-sized
                import psutil                                                    # temporaries per worker is a generous bound
                per_worker = 40 * rows * req['dim'] * 4 * req.get('mem_factor', 1) + (600 << 20)
                workers = max(1, min(workers, int(0.5 * psutil.virtual_memory().available / per_worker)))
            except Exception:
                pass
            procs = [subprocess.Popen(base + ['run', str(th), str(rows), str(passes), str(i * th)], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True, bufsize=1) for i in range(workers)]
            for p_ in procs:
                assert p_.stdout.readline().strip() == 'READY'
            print(f'[cpu pool] {workers} workers x {th} threads ready ({time.perf_counter() - t_req:.1f} s): {passes} passes x {rows} rows each',
                  file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            for p_ in procs:
                p_.stdin.write('go\n')
                p_.stdin.flush()
            outs = [json.loads(p_.stdout.readline()) for p_ in procs]
            wall = time.perf_counter() - t0
            for p_ in procs:
                p_.wait(timeout=60)
            print(f'[cpu pool] done: {wall:.1f} s wall', file=sys.stderr, flush=True)
            ans = {'ok': True, 'workers': workers, 'threads': th, 'rows_per_worker': rows * passes, 'rows_per_pass': rows, 'passes': passes,
                   'wall_s': wall, 'worker_s': [o['dt'] for o in outs], 'value': workers * rows * passes / wall,
                   'one_process_value': rate1, 'avail': avail}
        except Exception as e:                                                  # noqa: BLE001 -- answer, never die
            ans = {'ok': False, 'error': repr(e)[:300]}
        sys.stdout.write(json.dumps(ans) + '\n')
        sys.stdout.flush()
    return 0


def start_cpu_pool():
    """The pool process (see cpu_pool), or None when it cannot be started (e.g. under a profiler whose preloaded library has
    already initialised the GPU: cpu_baseline then falls back to one in-process oracle)."""
    if 'rocprof' in os.environ.get('LD_PRELOAD', '') or os.environ.get('ROCPROFILER_REGISTER_FORCE_LOAD'):
        return None
    try:
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-pool'], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                text=True, bufsize=1, start_new_session=True)       # own process group: stop_cpu_pool can end all of it
    except Exception:
        return None


def stop_cpu_pool(pool, hard=False):
    """End the pool (politely; `hard`: its whole process group at once -- a request that overran its deadline)."""
    import signal
    if pool is None or pool.poll() is not None:
        return
    try:
        if hard:
            os.killpg(pool.pid, signal.SIGKILL)
        else:
            pool.stdin.write('{"quit": true}\n')
            pool.stdin.flush()
        pool.wait(timeout=10)
    except Exception:
        try:
            os.killpg(pool.pid, signal.SIGKILL)
        except Exception:
            pass


def cpu_baseline(desc, state, dim=DIM, budget_s=10.0, max_rows=1 << 21, what='cfg2', pool=None):
    """The oracle (torch-CPU port of the reference's op sequence, incl. its double conditioner call; spline flows WITHOUT the
    reference's O(M^2) domain-check broadcast, SURVEY App. B Q1 -- with it the reference cannot run these sizes at all) timed on
    this box's host cores on a bounded sample.  One torch process stops scaling at a few dozen threads, so the all-core figure is
    floor(cores / best thread count) pinned oracle PROCESSES over disjoint row blocks, released together (`pool`: cpu_pool, started
    before this process touched the GPU); `value` = all rows / wall time from the release to the last worker's end, `cores` = threads x
    processes.  Without a pool: one process at its best thread count."""
    import tempfile
    import torch
    note = '' if what == 'cfg2' or what == 'cfg4' else ', O(M^2) domain-check broadcast of rational_quadratic_spline.py:167-178 omitted'
    avail = usable_cores()
    probe_rows = 1 << 16 if what == 'cfg2' else 1 << 12
    if pool is not None and pool.poll() is None:
        path = None
        try:
            fd_, path = tempfile.mkstemp(suffix='.pt', prefix='stribor_cpu_baseline_')
            os.close(fd_)
            torch.save({'desc': desc, 'state': {k: v.detach().cpu() for k, v in state.items()}, 'dim': dim}, path)
            pool.stdin.write(json.dumps({'spec': path, 'dim': dim, 'probe_rows': probe_rows, 'budget_s': budget_s,
                                         'rows_per_pass': min(max_rows, 1 << 18 if what == 'cfg2' else 1 << 14),
                                         'mem_factor': 1 if what != 'cfg3' else 24}) + '\n')
            pool.stdin.flush()
            # a deadline on the answer (probe + worker start-up + the sample, with a wide margin for a loaded box): a pool that
            # overruns it is ended with all its workers and this configuration falls back to the one-process figure
            import select
            deadline = 90.0 + 8.0 * budget_s
            ready, _, _ = select.select([pool.stdout], [], [], deadline)
            if not ready:
                stop_cpu_pool(pool, hard=True)
                raise TimeoutError(f'the worker pool gave no answer within {deadline:.0f} s')
            ans = json.loads(pool.stdout.readline())
        except Exception as e:                                                  # noqa: BLE001
            ans = {'ok': False, 'error': repr(e)[:300]}
        finally:
            if path and os.path.exists(path):
                os.unlink(path)
        if ans.get('ok'):
            return {'value': ans['value'], 'unit': 'samples/s', 'cores': ans['workers'] * ans['threads'], 'processes': ans['workers'],
                    'threads_per_process': ans['threads'], 'host_cores': os.cpu_count(), 'host_cores_available': avail,
                    'one_process_value': ans['one_process_value'], 'kind': 'port',
                    'sample': f'oracle.flow_log_prob (torch CPU fp32, reference op sequence incl. double conditioner call{note}): '
                              f'{ans["workers"]} pinned processes x {ans["threads"]} threads, each {ans["passes"]} passes over its own '
                              f'{ans["rows_per_pass"]} rows x {dim}, released together, {ans["wall_s"]:.2f} s wall'}
        pool_error = ans.get('error')
    else:
        pool_error = 'no worker pool (started under a profiler, pool start failed, or an earlier request overran its deadline)'
    from oracle import stribor_oracle as orc
    from stribor_amd.util import flowdesc as fd
    spec = fd.flow_spec(desc, state)
    cores, probe = _oracle_probe(spec, dim, probe_rows, avail, budget_s)
    with torch.no_grad():
        torch.set_num_threads(cores)
        rows = probe_rows
        while rows < max_rows and probe * (2 * rows / probe_rows) < budget_s:
            rows *= 2
        x = torch.randn(rows, dim, generator=torch.Generator().manual_seed(1234))
        t0 = time.perf_counter()
        orc.flow_log_prob(spec, x)
        dt = time.perf_counter() - t0
    return {'value': rows / dt, 'unit': 'samples/s', 'cores': cores, 'processes': 1, 'threads_per_process': cores,
            'host_cores': os.cpu_count(), 'host_cores_available': avail, 'kind': 'port', 'all_core_run_missing_because': pool_error,
            'sample': f'oracle.flow_log_prob (torch CPU fp32, reference op sequence incl. double conditioner call{note}) '
                      f'on {rows} rows x {dim}, 1 pass, {dt:.2f} s, ONE process'}


def load_profile(name):
    """A committed rocprofv3 summary (profiles/*.json), or {} when it was captured on a DIFFERENT build of the library than the
    one being timed (sx_build_id: sha256 over the library's sources): counter-derived fields are then reported as null
    rather than as observations of this run."""
    try:
        prof = json.load(open(os.path.join(ROOT, 'profiles', name)))
    except Exception:
        return {}
    try:
        from stribor_amd import _hip
        if prof.get('build_id') != _hip.build_id():
            return {'stale_profile': name, 'profile_build_id': prof.get('build_id'), 'library_build_id': _hip.build_id()}
    except Exception:
        return {}
    return prof


def time_extra_config(name, workload, flow, x, steps, warmup, flops_per_row, bytes_per_row, kernel, pmc, extra=None, exact=False):
    """One BASELINE configuration beside the headline: K steps of log_prob_sum over the resident batch, wall clock
    around barrier-free synchronised K steps plus HIP events over the same region for the kernel's launch period.
    exact: the v_mfma_f32_32x32x2_f32 arithmetic (stribor_amd.set_gemm_precision('exact')), priced against the fp32 MFMA peak."""
    import torch
    import stribor_amd as st
    out = torch.zeros(1, dtype=torch.float64, device=x.device)

    def step():
        out.zero_()
        flow.log_prob_sum(x, out)

    old_prec = st.set_gemm_precision('exact' if exact else 'fast')
    try:
        step()
        torch.cuda.synchronize()
        for _ in range(max(warmup, 3)):
            step()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:
        st.set_gemm_precision(old_prec)
    assert torch.isfinite(out).all(), name
    k_ms = ev0.elapsed_time(ev1) / steps
    rows = x.shape[0]
    tflops = flops_per_row * rows / (k_ms * 1e-3) / 1e12
    peak = PEAK_F32_MFMA_TFLOPS if exact else PEAK_F16_MFMA_TFLOPS
    roof = {'kernel': kernel, 'bound': 'mfma', 'achieved': tflops, 'peak': peak, 'unit': 'TFLOP/s',
            'frac': tflops / peak, 'traffic': pmc.get('hbm_bytes_per_launch'),
            'traffic_pmc_build_id': pmc.get('build_id'), 'traffic_stale_profile': pmc.get('stale_profile'), 'avg_kernel_ms': k_ms,
            'algorithmic_flops_per_launch': flops_per_row * rows, 'algorithmic_bytes_per_launch': bytes_per_row * rows,
            'hbm_frac_of_same_kernel': bytes_per_row * rows / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
    if exact:
        roof['gemm_arithmetic'] = 'v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, no operand range limit)'
    else:
        roof.update({'mfma_executed_tflops': 3 * tflops, 'frac_mfma_pipe_busy': 3 * tflops / PEAK_F16_MFMA_TFLOPS})
    if extra:
        roof.update(extra(k_ms, rows))
    return {'name': name, 'workload': workload, 'steps': steps, 'ms_per_step': elapsed / steps * 1e3,
            'value': rows * steps / elapsed, 'unit': 'samples/s', 'dtype': 'f32', 'precision': 'exact' if exact else 'fast',
            'log_prob_sum': out.item(), 'roofline': roof}


def extra_flow_entries(st, fd, dev, gen, steps):
    """Perf lines of the on-path families beyond the BASELINE configurations, 2^20 rows, D = 64:
      * the reference's flagship stack (test_normalizing_flow.py:13-35 without the CNF layer): affine coupling -> Flip -> Sigmoid ->
        cubic-spline coupling -> Logit, as one fused launch against the same flow evaluated layer by layer;
      * a NeuralFlow of 4 ContinuousAffineCoupling layers (coupling.py:98-213, flow.py:155-184): forward(x, t)."""
    import torch
    out = []
    rows = ROWS_PER_GPU

    def timed(fn, reps=steps):
        fn()
        torch.cuda.synchronize()
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    try:
        torch.manual_seed(0)
        D, H, K = 64, 64, 16
        stack = st.NormalizingFlow(st.UnitNormal(D), [
            st.Coupling(st.Affine(D, latent_net=st.net.MLP(D, [H], 2 * D)), mask='ordered_1'),
            st.Flip(dims=[-1]),
            st.Sigmoid(),
            st.Coupling(st.Spline(D, n_bins=K, latent_net=st.net.MLP(D, [H], D * (2 * K + 2)), spline_type='cubic'), mask='ordered_0'),
            st.Logit()]).to(dev)
        x = torch.randn(rows, D, device=dev, generator=gen)
        fused = stack._fused_program(True, D, 0, dev) is not None
        ms = timed(lambda: stack.log_prob(x))

        def layerwise():
            cur, acc = x, 0
            for f in reversed(stack.transforms):
                cur, l = f.inverse_and_log_det_jacobian(cur)
                acc = acc + l
            return stack.base_dist.log_prob(cur).unsqueeze(-1) + acc
        ms_l = timed(layerwise, max(3, steps // 4))
        out.append({'name': 'reference_stack', 'workload': 'affine coupling -> Flip -> Sigmoid -> cubic-spline coupling (K=16) -> Logit, D=64, '
                    '2^20 rows fp32, log_prob (stribor/test/test_normalizing_flow.py:13-35 without the CNF layer)',
                    'one_fused_launch': bool(fused), 'ms_per_step': ms, 'value': rows / (ms * 1e-3), 'unit': 'samples/s',
                    'ms_per_step_layer_by_layer': ms_l, 'speedup_vs_layer_by_layer': ms_l / ms})
        del stack, x
    except Exception as e:                                   # a perf line must not take the headline down
        out.append({'name': 'reference_stack', 'error': repr(e)})
    try:
        torch.manual_seed(0)
        D, H = 64, 64
        layers = [st.ContinuousAffineCoupling(latent_net=st.net.MLP(D + 1, [H], 2 * D), time_net=st.net.TimeTanh(2 * D),
                                              mask='ordered_0' if i % 2 == 0 else 'ordered_1') for i in range(4)]
        nf = st.NeuralFlow(layers).to(dev)
        x = torch.randn(rows, D, device=dev, generator=gen)
        t = torch.rand(rows, 1, device=dev, generator=gen)
        ms = timed(lambda: nf(x, t), max(3, steps // 2))
        out.append({'name': 'neural_flow', 'workload': 'NeuralFlow of 4 ContinuousAffineCoupling(MLP(65,[64],128), TimeTanh), D=64, 2^20 rows '
                    'fp32, forward(x, t) (flow.py:155-184, coupling.py:98-213)', 'ms_per_step': ms, 'value': rows / (ms * 1e-3),
                    'unit': 'samples/s', 'launches': 'see DESIGN 4.5'})
        del nf, x, t
    except Exception as e:
        out.append({'name': 'neural_flow', 'error': repr(e)})
    try:
        # spline couplings whose conditioner is wider than the one-launch program holds (hidden 160): the slab forward tier
        # (sx_rqs_slab_hidden + sx_rqs_slab_fwd per layer), 4 layers, 2^18 rows.  (Its hidden-64 neighbour -- one launch of the cfg-3
        # kernel on a smaller program -- is timed by tools/bench_cliffs.py, not here: its launches would be averaged into the cfg-3
        # kernel's counters by tools/profile_bench.sh.)
        torch.manual_seed(0)
        D, K, H, n = 64, 16, 160, 1 << 18
        x = torch.randn(n, D, device=dev, generator=gen)
        desc = [{'kind': 'coupling_rqs', 'dim': D, 'hidden': [H], 'mask': m, 'latent_dim': 0, 'n_bins': K, 'lower': -3, 'upper': 3}
                for m in ['ordered_right_half', 'ordered_left_half'] * 2]
        fl = fd.build_flow(st, desc, D).to(dev)
        ms = timed(lambda: fl.log_prob(x), max(3, steps // 4))
        out.append({'name': 'rqs_hidden160', 'workload': '4 rational-quadratic spline couplings (K=16), conditioner MLP(64,[160],1504), D=64, '
                    '2^18 rows fp32, log_prob (flows/spline.py:76-87 with a hidden layer beyond the one-launch tier)',
                    'tier': 'slab forward (2 launches per layer)', 'ms_per_step': ms, 'value': n / (ms * 1e-3), 'unit': 'samples/s',
                    'hidden64_neighbour': 'profiles/r05_cliffs.jsonl'})
        del fl, x
    except Exception as e:
        out.append({'name': 'rqs_hidden160', 'error': repr(e)})
    return out


def time_training_step(name, workload, flow, x, steps, note, optimizer=False):
    """`_time_training_step`, but a failing entry (e.g. a GemmRangeError of the fp16 x 3 arithmetic on some initialisation) is
    recorded as such instead of taking the headline line down with it."""
    import torch
    try:
        return _time_training_step(name, workload, flow, x, steps, note, optimizer)
    except Exception as e:
        torch.cuda.synchronize()
        try:
            import stribor_amd as st
            st.check_errors()
        except Exception:
            pass
        return {'name': name, 'workload': workload, 'error': repr(e)[:300], 'finite': False, 'ms_per_step': None}


def _time_training_step(name, workload, flow, x, steps, note, optimizer=False):
    """Training step beside the inference lines: forward + backward of loss = -log_prob(x).mean(), all
    parameter gradients; median over event-timed groups of 3 steps after warm-up.  optimizer: + an SGD step (lr = 0: the values
    stay, the parameter versions move, so every Linear of the flow is re-laid into MFMA fragments before the next forward -- what
    a training LOOP pays and the plain step does not)."""
    import torch
    opt = torch.optim.SGD(flow.parameters(), lr=0.0) if optimizer else None

    def step():
        for p_ in flow.parameters():
            p_.grad = None
        loss = -flow.log_prob(x).mean()
        loss.backward()
        if opt is not None:
            opt.step()
        return loss

    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    ts, inner = [], 3                      # events around groups of steps: a per-step sync would expose the launch ramp-up
    for _ in range(max(1, steps // inner)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            loss = step()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / inner)
    ts.sort()
    ms = ts[len(ts) // 2]
    finite = bool(torch.isfinite(loss).item()) and all(bool(torch.isfinite(p_.grad).all()) for p_ in flow.parameters())
    return {'name': name, 'workload': workload, 'steps': steps, 'ms_per_step': ms, 'value': x.shape[0] / (ms * 1e-3),
            'unit': 'samples/s', 'dtype': 'f32', 'finite': finite, 'note': note}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-pool':
        sys.exit(cpu_pool(sys.argv[2:]))
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-worker':
        sys.exit(cpu_worker(sys.argv[2:]))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the cfg 3 / cfg 4 lines (N = 1)')
    ap.add_argument('--no-training', action='store_true', help='skip the training-step entries (N = 1)')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help='collective backend of the N > 1 run.  nccl (= RCCL over xGMI) is the product path: one rank per GPU.  gloo '
                         'exists so that the WHOLE multi-rank control path (rank spawn, rendezvous, the async all-reduce ring, the '
                         'MAX over ranks of the elapsed time, rank 0\'s line) can run on a box with fewer GPUs than ranks: ranks share '
                         'GPUs round-robin and the numbers are a plumbing check, not a measurement')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    # the CPU baseline's worker pool starts HERE, before anything below touches the GPU (see cpu_pool)
    pool = start_cpu_pool() if args.gpus == 1 and 'WORLD_SIZE' not in os.environ and not args.no_cpu_baseline else None

    import torch
    import torch.distributed as dist

    import stribor_amd as st
    from stribor_amd import _hip
    from stribor_amd.sharded import ShardedLogProb
    from stribor_amd.util import flowdesc as fd

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev_index = 0
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dev_index = local_rank
            torch.cuda.set_device(dev_index)
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dev_index = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(dev_index)
            dist.init_process_group('gloo')
    dev = torch.device('cuda', dev_index)
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    rccl_ranks = dist.get_world_size() if world > 1 else 1

    torch.manual_seed(0)                                               # same weights on every rank
    desc = fd.cfg2_desc(LAYERS, DIM, HIDDEN)
    flow = fd.build_flow(st, desc, DIM)
    state = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(ROWS_PER_GPU, DIM, device=dev, generator=gen).bfloat16()   # resident before timing
    sharded = ShardedLogProb(flow)
    # A ring of result buffers: batch i's 8-byte all-reduce is enqueued asynchronously and overlaps batch i+1's kernel;
    # a buffer is reused only after the collective that owns it has been waited for (N = 1: no collective at all).
    RING = 4
    outs = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(RING)]
    pending = [None] * RING
    counter = [0]

    def step():
        i = counter[0] % RING
        counter[0] += 1
        if pending[i] is not None:
            pending[i].wait()
        pending[i] = sharded.log_prob_sum_async(x, outs[i])
        return pending[i]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # Untimed pre-run (first launch: weight packing, module load, counters; then a FIXED number of steps -- the same
        # on every rank, each step carries a collective -- so that the device leaves its idle power state) before the W
        # warm-up steps of the contract.
        step()
        torch.cuda.synchronize()
        for _ in range(25):
            for _ in range(20):
                step()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()                      # HIP events on the launch stream, around exactly the K timed steps
        for _ in range(args.steps):
            last = step()
        ev1.record()
        for p in pending:                # every outstanding collective completes inside the timed region
            if p is not None:
                p.wait()
        barrier()
        elapsed = time.perf_counter() - t0
    total = last.wait()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    assert torch.isfinite(total).all()
    total_value = total.item()

    result = None
    if rank == 0:
        rows_total = ROWS_PER_GPU * world * args.steps
        # dominant kernel: the fused flow kernel, one launch per step; average launch period over the timed region itself
        # (HIP events ev0 / ev1; each step also launches a 4.6 us zero-fill of the 8-byte result and, since round 6, the exact
        # redo pass of sx_flow_run2 -- on this data an empty list: its workgroups read one word and leave); the median of grouped
        # launches afterwards is kept beside it
        k_avg_ms = ev0.elapsed_time(ev1) / args.steps
        with torch.no_grad():
            _, k_med_ms = event_ms(lambda: flow.log_prob_sum(x, outs[0]), 10)
        achieved_tflops = FLOPS_PER_ROW * ROWS_PER_GPU / (k_avg_ms * 1e-3) / 1e12
        peak = PEAK_F16_MFMA_TFLOPS
        pmc = load_profile('pmc_cfg2.json')
        sq = load_profile('sq_cfg2.json')
        # standalone element-wise affine coupling kernel (params precomputed in HBM), HBM roofline
        from stribor_amd.flows.affine import run_affine_kernel
        params = torch.randn(ROWS_PER_GPU, DIM, device=dev) * 0.1
        e_avg_ms, _ = event_ms(lambda: run_affine_kernel(x, params, DIM, None, 0, DIM // 2, True, True, True, -1.0), 10)
        del params
        e_gbs = ELEMWISE_BYTES_PER_ROW * ROWS_PER_GPU / (e_avg_ms * 1e-3) / 1e9
        result = {
            'metric': 'log_prob samples/sec, D=64 8-layer affine-coupling',
            'value': rows_total / elapsed,
            'unit': 'samples/s',
            'n_gpus': world,
            'rccl_ranks': rccl_ranks if args.backend == 'nccl' else 0,
            'collective_backend': args.backend if world > 1 else None,
            'collective_ranks': rccl_ranks,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'cfg2: D=64, 8 alternating st.Affine coupling layers (MLP hidden 64, Tanh), '
                                   'batch 2^20 per GPU, x stored bf16, fp32 arithmetic, UnitNormal base, fp64 batch sum',
                       'rows_per_gpu': ROWS_PER_GPU, 'dim': DIM, 'layers': LAYERS, 'hidden': HIDDEN,
                       'x_storage': 'bf16', 'parallelism': f'batch-sharded x{world}, one 8-byte all-reduce per step'},
            'roofline': {'kernel': 'flow_fused_kernel<NS=1,TX=2,HT=2,MODE=5> (pure split-coupling program, 8-wave workgroups)', 'bound': 'mfma',
                         'achieved': achieved_tflops, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved_tflops / peak,
                         'traffic': pmc.get('flow_fused_kernel', {}).get('hbm_bytes_per_launch'),
                         'traffic_pmc_build_id': pmc.get('build_id'), 'traffic_stale_profile': pmc.get('stale_profile'),
                         'library_build_id': _hip.build_id(),
                         'avg_kernel_ms': k_avg_ms, 'median_kernel_ms': k_med_ms,
                         'gemm_arithmetic': 'fp16 x 3 split on v_mfma_f32_32x32x16_f16, fp32 accumulate '
                                            '(3 MFMA products per algorithmic product)',
                         'mfma_executed_tflops': achieved_tflops * 3,
                         'frac_mfma_pipe_busy': achieved_tflops * 3 / peak,
                         'frac_of_exact_fp32_mfma_peak': achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                         'binding_resource': 'VALU issue (tanh/exp transcendentals + fp16 operand splitting); '
                                             'neither HBM (3 % of peak) nor the matrix pipe binds',
                         'valu_issue_busy_frac_pmc': sq.get('valu_issue_busy_frac'),
                         'mfma_pipe_busy_frac_pmc': sq.get('mfma_pipe_busy_frac'),
                         'effective_clock_ghz_pmc': sq.get('effective_clock_ghz'),
                         'sq_pmc_build_id': sq.get('build_id'), 'sq_stale_profile': sq.get('stale_profile'),
                         'algorithmic_flops_per_launch': FLOPS_PER_ROW * ROWS_PER_GPU,
                         'algorithmic_bytes_per_launch': BYTES_PER_ROW * ROWS_PER_GPU,
                         'hbm_frac_of_same_kernel': BYTES_PER_ROW * ROWS_PER_GPU / (k_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            'roofline_elementwise': {'kernel': 'affine_coupling_vec_kernel<bf16,reverse> (16 B per lane)', 'bound': 'hbm',
                                     'achieved': e_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': e_gbs / PEAK_HBM_GBS,
                                     'traffic': (pmc.get('affine_coupling_vec_kernel') or pmc.get('affine_coupling_vec4_kernel') or {}).get('hbm_bytes_per_launch'),
                                     'traffic_pmc_build_id': pmc.get('build_id'),
                                     'avg_kernel_ms': e_avg_ms,
                                     'algorithmic_bytes_per_launch': ELEMWISE_BYTES_PER_ROW * ROWS_PER_GPU},
            'log_prob_sum': total_value,
        }
        if world == 1 and not args.no_extra_configs:
            # BASELINE cfg 3 / cfg 4 at their full 2^20 rows, fp32 storage, same process, after the headline; beside each its own
            # CPU baseline (the oracle on this box's cores, bounded sample), and cfg 2 once more in the exact-fp32 arithmetic
            extra_steps = max(5, min(args.steps, 20))
            cfgs = []
            with torch.no_grad():
                cfgs.append(time_extra_config(
                    'cfg2_exact', 'cfg2 (headline workload, x stored bf16) in the exact arithmetic: set_gemm_precision("exact")',
                    flow, x, extra_steps, args.warmup, FLOPS_PER_ROW, BYTES_PER_ROW,
                    'flow_fused_kernel<NS=1,TX=2,HT=2,MODE=5>, sx_f32x build (v_mfma_f32_32x32x2_f32)', {}, exact=True))
                del x
                torch.manual_seed(0)
                f3 = fd.build_flow(st, fd.cfg3_desc(), 64)
                s3 = {k_: v_.clone() for k_, v_ in f3.state_dict().items()}
                f3 = f3.to(dev)
                x3 = torch.randn(ROWS_PER_GPU, 64, device=dev, generator=gen)
                cfgs.append(time_extra_config(
                    'cfg3', 'cfg3: D=64, 8 rational-quadratic st.Spline couplings, K=16 bins on [-3, 3] (MLP hidden 64), '
                    'batch 2^20, fp32', f3, x3, extra_steps, args.warmup, CFG3_FLOPS_PER_ROW, CFG3_BYTES_PER_ROW,
                    'flow_fused_kernel<NS=1,TX=2,HT=2,MODE=3> (spline phases: parameters only in registers)',
                    load_profile('pmc_cfg3.json')))
                if not args.no_cpu_baseline:
                    cfgs[-1]['cpu_baseline'] = cpu_baseline(fd.cfg3_desc(), s3, 64, budget_s=4.0, max_rows=1 << 16, what='cfg3', pool=pool)
                del f3, x3
                torch.manual_seed(0)
                f4 = fd.build_flow(st, fd.cfg4_desc(), 128)
                s4 = {k_: v_.clone() for k_, v_ in f4.state_dict().items()}
                f4 = f4.to(dev)
                x4 = torch.randn(ROWS_PER_GPU, 128, device=dev, generator=gen)
                cfgs.append(time_extra_config(
                    'cfg4', 'cfg4: D=128, 4 x [st.AffineLU, affine coupling, st.MatrixExponential (t=1), affine coupling] '
                    '(MLP hidden 64), batch 2^20, fp32', f4, x4, extra_steps, args.warmup, CFG4_FLOPS_PER_ROW,
                    CFG4_BYTES_PER_ROW, 'flow_fused_kernel<NS=1,TX=4,HT=2,MODE=7> (dense linear layers + split couplings)',
                    load_profile('pmc_cfg4.json'),
                    extra=lambda k_ms, rows: {'frac_collapsed': CFG4_FLOPS_PER_ROW_COLLAPSED * rows / (k_ms * 1e-3) / 1e12
                                              / PEAK_F16_MFMA_TFLOPS,
                                              'note': 'achieved counts the reference\'s two triangular products per '
                                                      'MatrixExponential (SURVEY 8(d): 589,824 flop/row); the kernel '
                                                      'runs the collapsed single matrix (458,752): frac_collapsed'}))
                if not args.no_cpu_baseline:
                    cfgs[-1]['cpu_baseline'] = cpu_baseline(fd.cfg4_desc(), s4, 128, budget_s=4.0, max_rows=1 << 16, what='cfg4', pool=pool)
                del f4, x4
                cfgs += extra_flow_entries(st, fd, dev, gen, extra_steps)
            result['configs'] = cfgs
            # training steps (forward + backward) of the trainable BASELINE families, same process
            tr = []
            if not args.no_training:
                torch.manual_seed(0)
                f2 = fd.build_flow(st, fd.cfg2_desc(), 64).to(dev)
                x2t = torch.randn(ROWS_PER_GPU, 64, device=dev, generator=gen)
                tr.append(time_training_step('cfg2_train', 'cfg2 flow, 2^20 rows fp32: loss = -log_prob.mean(), backward to every '
                                             'parameter', f2, x2t, 12,
                                             'layer-major backward: weight gradients contracted in the kernel (DESIGN 4.3)'))
                del f2, x2t
                torch.manual_seed(0)
                f3 = fd.build_flow(st, fd.cfg3_desc(), 64).to(dev)
                x3t = torch.randn(ROWS_PER_GPU // 4, 64, device=dev, generator=gen)
                tr.append(time_training_step('cfg3_train', 'cfg3 flow, 2^18 rows fp32: loss = -log_prob.mean(), backward to every '
                                             'parameter', f3, x3t, 12,
                                             'spline backward fused with the last conditioner layer, no [N, 1504] parameter tensor '
                                             '(sx_rqs_slab_bwd, DESIGN 4.3.1); HBM bytes per step in profiles/pmc_training_cfg3_fused.json'))
                tr.append(time_training_step('cfg3_train_sgd', 'cfg3 flow, 2^18 rows fp32: the same step + an SGD optimizer step (every weight '
                                             're-packed into fragments before the next forward)', f3, x3t, 9,
                                             'one sx_pack_linear_batch launch per re-packed program (104 + 24 single pack launches per step before '
                                             'round 6: 10.6 -> 9.3 ms on one box, profiles/r06_ab_head_training_optim.txt)', optimizer=True))
                del f3
                pt = load_profile('pmc_training_cfg3_fused.json')
                if pt and not pt.get('stale_profile'):
                    tr[-2]['hbm_MB_per_step_pmc'] = {'read': pt.get('hbm_read_MB_per_step'), 'written': pt.get('hbm_write_MB_per_step'),
                                                     'source': 'profiles/pmc_training_cfg3_fused.json (builder-run rocprofv3 passes, same build id)'}
                torch.manual_seed(0)
                cub = [dict(d_, spline_type='cubic') for d_ in fd.cfg3_desc()]
                f3c = fd.build_flow(st, cub, 64).to(dev)
                tr.append(time_training_step('cfg3_cubic_train', 'cfg3 shape with spline_type="cubic" (the reference\'s default), 2^18 rows '
                                             'fp32: loss = -log_prob.mean(), backward to every parameter', f3c, x3t, 9,
                                             'fused MODE 12 forward + cubic slab backward (DESIGN 4.3.1)'))
                del f3c, x3t
                torch.manual_seed(0)
                f4 = fd.build_flow(st, fd.cfg4_desc(), 128).to(dev)
                for rows_ in (ROWS_PER_GPU // 4, ROWS_PER_GPU):
                    x4t = torch.randn(rows_, 128, device=dev, generator=gen)
                    tr.append(time_training_step(f'cfg4_train_2^{rows_.bit_length() - 1}', f'cfg4 flow, {rows_} rows fp32: loss = -log_prob.mean(), '
                                                 'backward to every parameter', f4, x4t, 9 if rows_ < ROWS_PER_GPU else 6,
                                                 'single-launch backward program on 4 + 4 tiles (SX_STEP_LINEAR_BWD / COUPLING_AFFINE_BWD_A / _B), '
                                                 'weight gradients by sx_wgrad on the stored factors, D x D parameter algebra in batched fp64 '
                                                 'torch ops: no library GEMM (DESIGN 4.3.2)'))
                    del x4t
                del f4
            if not args.no_training:
                result['training'] = tr
                for t_ in tr:
                    if t_.get('ms_per_step') is not None:
                        result['train_ms_' + t_['name']] = t_['ms_per_step']
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(desc, state, pool=pool)
        # LAST keys of the line (the driver's record keeps the tail of stdout): first-level scalars of the other single-GPU BASELINE
        # configurations of this run -- rows/s, ms per step and fraction of the dense fp16 MFMA peak (cfg2_exact: of the fp32 MFMA peak)
        for c_ in result.get('configs', []):
            if c_.get('name') in ('cfg2_exact', 'cfg3', 'cfg4') and 'value' in c_:
                result['value_' + c_['name']] = float('%.4g' % c_['value'])
                result['ms_per_step_' + c_['name']] = round(c_['ms_per_step'], 4)
                result['roofline_frac_' + c_['name']] = round(c_['roofline']['frac'], 4)
    stop_cpu_pool(pool)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
