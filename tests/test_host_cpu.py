"""CPU: host logic of the product — library ABI, constructor / state_dict surface, masks, planner."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import flowdesc as fd
from goldens import Golden

import stribor_amd as st
from stribor_amd import _hip
from stribor_amd.fused import ProgramBuilder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/stribor_hip.h declares."""
    hdr = open(os.path.join(ROOT, 'include', 'stribor_hip.h')).read()
    declared = set(re.findall(r'\b(sx_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'sx_step', 'sx_program'}
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(_hip.EXPORTS) == declared
    assert _hip.lib().sx_abi_version() == _hip.SX_ABI_VERSION == 3
    assert _hip.lib().sx_packed_linear_floats(2, 1) == _hip.packed_linear_floats(2, 1) == 2 * 1024 + 64


def test_struct_layout_matches_header():
    assert ctypes.sizeof(_hip.sx_step) == 48
    assert ctypes.sizeof(_hip.sx_program) == 32 + 48 * _hip.SX_MAX_STEPS


def test_ctypes_records_match_the_header_as_a_c_compiler_lays_it_out():
    """include/stribor_hip.h compiled as plain C (gcc): size and every field offset of the structs a binding fills -- sx_step,
    sx_program, and the device-table records of sx_pack_linear_batch / sx_wgrad_reduce_batch (a mismatch there would not fail a
    call: the kernel would read shifted pointers) -- against the ctypes mirrors in stribor_amd/_hip.py."""
    import subprocess
    import tempfile
    structs = {'sx_step': _hip.sx_step, 'sx_program': _hip.sx_program, 'sx_pack_job': _hip.sx_pack_job, 'sx_reduce_job': _hip.sx_reduce_job}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "stribor_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append('  printf("%s %%zu", sizeof(%s));' % (name, name))
        for field, _ in cls._fields_:
            lines.append('  printf(" %%zu", offsetof(%s, %s));' % (name, field))
        lines.append('  printf("\\n");')
    lines += ['  return 0;', '}']
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, 'layout.c')
        with open(src, 'w') as f:
            f.write('\n'.join(lines) + '\n')
        subprocess.run(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'), src, '-o', os.path.join(td, 'layout')], check=True)
        out = subprocess.run([os.path.join(td, 'layout')], check=True, capture_output=True, text=True).stdout
    seen = {}
    for line in out.strip().splitlines():
        name, *nums = line.split()
        seen[name] = [int(v) for v in nums]
    for name, cls in structs.items():
        want = [ctypes.sizeof(cls)] + [getattr(cls, field).offset for field, _ in cls._fields_]
        assert seen[name] == want, (name, seen[name], want)


def test_no_cpu_fallback():
    f = st.NormalizingFlow(st.UnitNormal(2), [st.Affine(2)])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        f.log_prob(torch.randn(3, 2))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        st.Permute(4)(torch.randn(3, 4))


def test_masks_exact():
    """stribor/test/test_mask.py:4-30 + captured vectors."""
    g = Golden('f2_masks')
    for key in g.arrays:
        name, d = key.split('/')
        assert torch.equal(st.util.get_mask(name)(int(d)), g.t(key)), key
    with pytest.raises(NotImplementedError):
        st.util.get_mask('nope')
    m = st.util.get_mask('random_half')(10)
    assert m.sum() == 5 and set(m.tolist()) == {0.0, 1.0}


@pytest.mark.parametrize('fixture,case', [('f3_cfg1', 'cfg1'), ('f4_cfg2', 'cfg2'), ('f7_permute', 'mixed')])
def test_state_dict_keys_match_reference(fixture, case):
    g = Golden(fixture)
    m = g.meta[case]
    flow = fd.build_flow(st, m['desc'], m['dim'])
    ref_state = g.state(case)
    assert set(flow.state_dict().keys()) == set(ref_state.keys())
    for k, v in flow.state_dict().items():
        assert tuple(v.shape) == tuple(ref_state[k].shape), k
    flow.load_state_dict(ref_state)
    # a checkpoint written by the reference has no permutation entry (quirk Q7): still loads
    flow.load_state_dict({k: v for k, v in ref_state.items() if 'permutation' not in k})


def test_constructor_surface():
    """Signatures of SURVEY 8(b): keyword-only Affine args, Spline defaults, error types."""
    with pytest.raises(TypeError):
        st.Affine(2, st.net.MLP(2, [4], 4))                       # latent_net is keyword-only (affine.py:31-35)
    with pytest.raises(ValueError):
        st.Spline(2, 3, spline_type='linear')                     # spline.py:63
    spc = st.Spline(2, 3)                                         # default spline_type is 'cubic' (spline.py:46)
    assert spc.spline_type == 'cubic' and spc.derivative.shape == (2, 2) and spc.params_per_element == 8
    with pytest.raises(AssertionError):
        st.Affine(2, scale=torch.tensor([1.0, -1.0]), shift=torch.zeros(2))    # affine.py:55
    mlp = st.net.MLP(3, [5, 7], 4)
    assert list(mlp.state_dict().keys()) == ['net.0.weight', 'net.0.bias', 'net.2.weight', 'net.2.bias',
                                             'net.4.weight', 'net.4.bias']
    assert torch.all(mlp.net[4].bias == 0)                        # mlp.py:53
    sp = st.Spline(4, 5, spline_type='quadratic')
    assert set(sp.state_dict()) == {'width', 'height', 'derivative'} and sp.derivative.shape == (4, 4)


def test_default_init_matches_reference_rng_stream():
    """Same construction order under the same seed -> same weights as the reference (fixture F4)."""
    g = Golden('f4_cfg2')
    torch.manual_seed(g.meta['cfg2']['seed'])
    flow = fd.build_flow(st, g.meta['cfg2']['desc'], 64)
    for k, v in flow.state_dict().items():
        assert torch.equal(v, g.t('cfg2/state/' + k)), k


def test_planner_prunes_ordered_and_parity_masks():
    def plan(masks, dim=64, hidden=64, perm_after=None):
        torch.manual_seed(0)
        layers = []
        for i, mk in enumerate(masks):
            layers.append(st.Coupling(st.Affine(dim, latent_net=st.net.MLP(dim, [hidden], 2 * dim)), mask=mk))
            if perm_after is not None and i == perm_after:
                layers.append(st.Permute(dim))
        b = ProgramBuilder(dim, 0, hidden)
        b.choose_layout(layers[0]._plan_first_mask(dim))
        for f in layers:
            assert f._plan(b, False, 1.0)
        return b

    b = plan(['ordered_right_half', 'ordered_left_half'] * 2)
    assert [(s['c0'], s['ct'], s['t0'], s['tt']) for s in b.steps] == [(1, 1, 0, 1), (0, 1, 1, 1)] * 2
    assert np.array_equal(b.col_of_slot, np.arange(64))            # identity layout -> vector loads
    b = plan(['parity_even', 'parity_odd'])
    assert [(s['ct'], s['tt']) for s in b.steps] == [(1, 1), (1, 1)]          # pruned via slot layout
    assert sorted(b.col_of_slot[:32]) == list(range(1, 64, 2))
    b = plan(['ordered_right_half', 'parity_even'])
    assert [(s['ct'], s['tt']) for s in b.steps] == [(1, 1), (2, 2)]          # second one runs dense
    b = plan(['ordered_right_half', 'ordered_left_half'], perm_after=0)
    assert b.steps[1]['ct'] == 2                                               # random permutation -> dense
    # blobs are 1 KiB aligned and large enough
    for s in b.steps:
        assert s['blob_off'] % 256 == 0 and s['blob_floats'] % 256 == 0


def test_planner_permutation_relabelling_matches_gather():
    """Folding Permute/Flip into slot labels == explicit gathers (permute.py:71,75), both directions."""
    rng = np.random.default_rng(0)
    for reverse in (False, True):
        b = ProgramBuilder(10, 0, 32)
        x = rng.standard_normal(10)
        ref = x.copy()
        for _ in range(3):
            perm = rng.permutation(10)
            inv = np.argsort(perm)
            b.add_permutation(perm, reverse)
            ref = ref[inv] if reverse else ref[perm]
        state = np.zeros(32)
        state[:10] = x                                   # slots never move
        out = np.zeros(10)
        for p, c in enumerate(b.col_of_slot):
            if c >= 0:
                out[c] = state[p]
        assert np.array_equal(out, ref)


def test_tile_limits_raise():
    with pytest.raises(NotImplementedError):
        ProgramBuilder(300, 0, 32)
    # round 4: hidden layers beyond four tiles plan as chunk steps (affine couplings, Tanh, split masks); everything else still refuses
    b = ProgramBuilder(64, 0, 300)
    assert b.h_tiles == 4
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(64, 300), torch.nn.Linear(300, 128)
    m = np.zeros(64)
    m[32:] = 1
    b.add_coupling_affine(lin1.weight, lin1.bias, lin2.weight, lin2.bias, m, _hip.ACT_CODES['Tanh'], True, -1.0, 300)
    assert [s['kind'] for s in b.steps] == [_hip.STEP_COUPLING_AFFINE_HC] * 3 and [s['pad_'] for s in b.steps] == [1, 0, 2]
    with pytest.raises(NotImplementedError):
        ProgramBuilder(64, 0, 300).add_coupling_affine(lin1.weight, lin1.bias, lin2.weight, lin2.bias, m, _hip.ACT_CODES['ReLU'], True, -1.0, 300)
    with pytest.raises(NotImplementedError):
        ProgramBuilder(64, 0, 300).add_coupling_rqs(lin1.weight, lin1.bias, lin2.weight, lin2.bias, m, True, -1.0, 300, 4, -1, 1, -1, 1)
    # (found by tools/fuzz_train.py --infer --fat in round 4: a builder for a 150-wide hidden layer carries the chunk width in
    #  h_tiles, and an MLP program of a two-hidden-layer conditioner indexed past it instead of leaving the layer to the next tier)
    lin_a, lin_b, lin_c = torch.nn.Linear(64, 150), torch.nn.Linear(150, 69), torch.nn.Linear(69, 94)
    with pytest.raises(NotImplementedError):
        ProgramBuilder(64, 0, 150).add_mlp([(lin_a.weight, lin_a.bias), (lin_b.weight, lin_b.bias), (lin_c.weight, lin_c.bias)],
                                           _hip.ACT_CODES['Tanh'], None, np.arange(94))
    bb = ProgramBuilder(64, 0, 150, min_x_tiles=1)
    bb.enable_adjoint_tiles()
    with pytest.raises(NotImplementedError):
        bb.add_coupling_affine_bwd(lin_a.weight, lin_a.bias, torch.nn.Linear(150, 128).weight, torch.zeros(128), m, 150, 0)


def test_planner_spline_flow_and_mixed_fallback():
    """cfg 3 plans into 8 x (1 hidden + 12 phase) steps; a flow mixing spline and affine couplings is not fused."""
    torch.manual_seed(0)
    flow = fd.build_flow(st, fd.cfg3_desc(), 64)
    b = ProgramBuilder(64, 0, 64)
    order = list(reversed(flow.transforms))
    b.choose_layout(order[0]._plan_first_mask(64))
    for f in order:
        assert f._plan(b, True, -1.0)
    assert len(b.steps) == 8 * 13
    phases = [s for s in b.steps if s['kind'] == _hip.STEP_RQS_PHASE]
    assert all(s['tt'] == 16 and s['pad_'] == -1 for s in phases)            # 16 bins, all 32 slots live
    assert [s['ct'] for s in phases[:3]] == [0, 1, 2]
    assert sum(1 for s in phases if s['ldj_scale'] != 0) == 8 * 4            # only the evaluate phases add log-det
    # round 3: a flow mixing spline and affine couplings is ONE mixed program (kernel MODE 14)
    mixed = st.NormalizingFlow(st.UnitNormal(64), [flow.transforms[0], fd.build_transform(st, fd.cfg2_desc(1)[0]), st.Sigmoid()])
    prog = mixed._build_fused(True, 64, 0, torch.device('cpu'))
    assert prog is not None
    kinds = [prog.prog.steps[i].kind for i in range(prog.prog.n_steps)]
    assert kinds[0] == _hip.STEP_POINTWISE and _hip.STEP_COUPLING_AFFINE in kinds and _hip.STEP_RQS_PHASE in kinds
    assert prog.prog.steps[0].act == 2                                           # Sigmoid's inverse pass runs the LOGIT kind
    # dense layers and Cumsum stay out of fused spline programs: layer by layer
    assert st.NormalizingFlow(st.UnitNormal(64), [flow.transforms[0], st.AffineLU(64)])._build_fused(True, 64, 0, torch.device('cpu')) is None
    assert st.NormalizingFlow(st.UnitNormal(64), [flow.transforms[0], st.Cumsum(-1)])._build_fused(True, 64, 0, torch.device('cpu')) is None


def test_planner_cubic_spline_flow_and_spline_mix_rules():
    """Round 2: cubic-spline couplings plan into the same hidden + 12-phase layout with act = 1 (kernel MODE 12 / 13); the planner
    sends the mixes the pure spline kernel variants cannot run -- quadratic with cubic splines, any spline with a two-hidden-layer
    affine coupling (which the spline kernel variant used to skip silently, tools/fuzz_train.py --infer) -- to the mixed program
    (kernel MODE 14, round 3) instead of producing a wrong fused program."""
    torch.manual_seed(0)
    dev = torch.device('cpu')
    cubic = [dict(d, spline_type='cubic') for d in fd.cfg3_desc(2)]
    flow = fd.build_flow(st, cubic, 64)
    b = ProgramBuilder(64, 0, 64)
    order = list(reversed(flow.transforms))
    b.choose_layout(order[0]._plan_first_mask(64))
    for f in order:
        assert f._plan(b, True, -1.0)
    phases = [s for s in b.steps if s['kind'] == _hip.STEP_RQS_PHASE]
    assert len(b.steps) == 2 * 13 and all(s['act'] == 1 and s['tt'] == 16 for s in phases)
    assert flow._build_fused(True, 64, 0, dev) is not None
    quad = fd.build_flow(st, fd.cfg3_desc(1), 64).transforms[0]
    assert st.NormalizingFlow(st.UnitNormal(64), [flow.transforms[0], quad])._build_fused(True, 64, 0, dev) is not None     # round 3: mixed program
    deep_affine = fd.build_transform(st, {'kind': 'coupling_affine', 'dim': 64, 'hidden': [48, 40], 'mask': 'ordered_left_half',
                                          'latent_dim': 0})
    assert st.NormalizingFlow(st.UnitNormal(64), [deep_affine])._build_fused(True, 64, 0, dev) is not None
    for spline in (quad, flow.transforms[0]):          # round 3: the mixed program (kernel MODE 14) carries deep conditioners too
        assert st.NormalizingFlow(st.UnitNormal(64), [deep_affine, spline])._build_fused(True, 64, 0, dev) is not None
    wide = fd.build_transform(st, {'kind': 'coupling_rqs', 'dim': 64, 'hidden': [64], 'n_bins': 20, 'lower': -3, 'upper': 3,
                                   'mask': 'ordered_left_half', 'latent_dim': 0, 'spline_type': 'cubic'})
    assert st.NormalizingFlow(st.UnitNormal(64), [wide])._build_fused(True, 64, 0, dev) is None        # n_bins > 16: layer-wise tier


def test_wide_mlp_program_chunks_write_their_own_windows():
    """An MLP program keeps its output-tile index in 8 bits: a 511-tile output is split over launches whose tile indices
    restart at 0, each with the column offset of its window (round 2: indices past 255 used to wrap on the device)."""
    net = st.net.MLP(20, [48], 8192 + 2 * 4032 + 77)
    progs = net._program(torch.device('cpu'))
    assert len(progs) >= 5
    col = 0
    for pr in progs:
        outs = [pr.prog.steps[i] for i in range(pr.prog.n_steps) if pr.prog.steps[i].kind == _hip.STEP_MLP_OUT_TILE]
        assert [o.t0 for o in outs] == list(range(len(outs))) and len(outs) < 256
        assert pr.mlp_col0 == col and pr.mlp_out_dim == min(32 * len(outs), net.out_dim - col)
        col += 32 * len(outs)
    assert col >= net.out_dim


# ---- round 2: program-cache validity (host logic only; no GPU) -----------------------------------------------------
def test_program_cache_epoch_guards_and_fingerprint():
    from stribor_amd.fused import ProgramCache, _STRUCT_EPOCH, bump_structure_epoch
    c = ProgramCache()
    built = []

    def build():
        built.append(1)
        return len(built)

    g = torch.arange(4)
    assert c.get('k', build, [g], ('a',)) == 1
    assert c.get('k', build, [g], ('a',)) == 1            # hit
    g.add_(1)                                             # in-place change of a guard tensor (load_state_dict)
    assert c.get('k', build, [g], ('a',)) == 2
    assert c.get('k', build, [g], ('b',)) == 3            # owner fingerprint changed (ModuleList edit)
    bump_structure_epoch()
    assert c.get('k', build, [g], ('b',)) == 4
    assert c.get('k', build, [g], ('b',)) == 4
    assert _STRUCT_EPOCH[0] > 0


def test_structure_epoch_bumps_on_module_edits():
    import torch.nn as nn
    from stribor_amd.fused import _STRUCT_EPOCH
    net = st.net.MLP(4, [8], 8)
    cpl = st.Coupling(st.Affine(4, latent_net=net), mask='ordered_left_half')
    flow = st.NormalizingFlow(st.UnitNormal(4), [cpl, st.Permute(4)])
    e = _STRUCT_EPOCH[0]
    flow.eval(); flow.train()                             # mode flips must NOT invalidate programs
    assert _STRUCT_EPOCH[0] == e
    net.net[0].weight = nn.Parameter(torch.zeros(8, 4))
    assert _STRUCT_EPOCH[0] > e; e = _STRUCT_EPOCH[0]
    net.net[2] = nn.Linear(8, 8)
    assert _STRUCT_EPOCH[0] > e; e = _STRUCT_EPOCH[0]
    cpl.transform.latent_net = st.net.MLP(4, [8], 8)
    assert _STRUCT_EPOCH[0] > e; e = _STRUCT_EPOCH[0]
    cpl.mask_func = st.util.get_mask('parity_odd')
    assert _STRUCT_EPOCH[0] > e
    assert list(cpl.mask_vector(4)) == [1.0, 0.0, 1.0, 0.0]            # mask cache follows the epoch
    # state_dict keys are the reference's (tracked Linear / Sequential subclasses change nothing there)
    assert sorted(net.state_dict()) == ['net.0.bias', 'net.0.weight', 'net.2.bias', 'net.2.weight']
    # guards: the permutation buffer; fingerprint: ids of the transforms
    assert flow._plan_guards()[0] is flow.transforms[1].permutation
    fp = flow._fingerprint()
    flow.transforms[0], flow.transforms[1] = flow.transforms[1], flow.transforms[0]
    assert flow._fingerprint() != fp


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 8` without a launcher: N fresh children under torch.distributed.run."""
    import argparse
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, 'run', fake_run)
    rc = bench.spawn_ranks(argparse.Namespace(gpus=8, steps=7, warmup=2, no_cpu_baseline=True))
    assert rc == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--gpus') + 1] == '8' and cmd[cmd.index('--steps') + 1] == '7'
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert '--backend' not in cmd                      # nccl (RCCL) is the default and the product path
    bench.spawn_ranks(argparse.Namespace(gpus=2, steps=3, warmup=1, no_cpu_baseline=True, backend='gloo'))
    assert seen['cmd'][seen['cmd'].index('--backend') + 1] == 'gloo'


@pytest.mark.parametrize('cubic', [False, True])
@pytest.mark.parametrize('n_live,K', [(32, 16), (5, 3), (1, 1), (7, 16), (2, 9)])
def test_slab_slot_rows_cover_every_parameter_row_once(n_live, K, cubic):
    """sx_rqs_slab_bwd's slot map (include/stribor_hip.h): every row of the selected last conditioner layer -- per transformed
    column K widths, K heights, K-1 derivatives (quadratic) or 2 (cubic) (spline.py:82-86) -- sits in exactly one slot, in the tile of its block and the
    lane half of its column; all other slots are padding."""
    import numpy as np
    from stribor_amd.flows.spline import slab_slot_rows
    rows = slab_slot_rows(n_live, K, cubic)
    P = 2 * K + 2 if cubic else 3 * K - 1
    n_third = 2 if cubic else K - 1
    n_slabs = (n_live + 1) // 2
    assert rows.shape == (n_slabs * 96,) and rows.dtype == np.int32
    used = rows[rows >= 0]
    assert sorted(used.tolist()) == list(range(n_live * P))
    for slot, r in enumerate(rows):
        if r < 0:
            continue
        s_, t, R = slot // 96, (slot % 96) // 32, slot % 32
        ci, off = divmod(int(r), P)
        assert ci == 2 * s_ + ((R >> 2) & 1)                         # column <-> lane half of the C fragment
        k = (R & 3) + 4 * (R >> 3)
        assert off == t * K + k and k < (n_third if t == 2 else K)   # block <-> tile, parameter <-> register


def test_hook_computed_tensors_do_not_invalidate_every_cached_program():
    """torch's spectral_norm recomputes `module.weight` (a plain tensor) in a forward pre-hook on every call; that
    assignment must not bump the process-wide structure epoch (every flow would re-plan and re-pack each step), while replacing a
    parameter, a buffer or a plain attribute still does."""
    from stribor_amd.fused import _STRUCT_EPOCH
    net = st.net.MLP(4, [8], 6, nn_linear_wrapper_func=torch.nn.utils.spectral_norm)
    wrapped = net.net[2]
    assert 'weight' not in wrapped._parameters and 'weight_orig' in wrapped._parameters
    e0 = _STRUCT_EPOCH[0]
    setattr(wrapped, 'weight', torch.randn(6, 8))                  # what the hook does
    assert _STRUCT_EPOCH[0] == e0
    wrapped.bias = torch.nn.Parameter(torch.zeros(6))              # a parameter replaced: plans must go
    assert _STRUCT_EPOCH[0] == e0 + 1
    p = st.Permute(5)
    e1 = _STRUCT_EPOCH[0]
    p.permutation = torch.randperm(5)                              # a registered buffer re-assigned
    assert _STRUCT_EPOCH[0] == e1 + 1
    c = st.Coupling(st.Affine(4, latent_net=st.net.MLP(4, [8], 8)), mask='ordered_0')
    e2 = _STRUCT_EPOCH[0]
    c.set_data = True                                              # a plain attribute the planner reads
    assert _STRUCT_EPOCH[0] == e2 + 1


def test_fused_kernel_stages_weights_with_mubuf_lds_dma():
    """Round 4: a FLAT-encoded LDS-DMA (global_load_lds) in flight makes hipcc's wait-count pass emit every later s_waitcnt as a full
    one, so no ds_read of the fused kernel could stay in flight across an MFMA group (cfg 3 -3.7 % from the switch alone).
    The fused kernel's headers and sx_wgrad must keep the MUBUF form (buffer_load ... lds)."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stribor_amd', 'csrc')
    for name in ('sx_flow_kernel.h', 'sx_flow_spline.h', 'sx_flow_bwd.h', 'sx_wgrad.hip'):
        src = open(os.path.join(root, name)).read()
        code = re.sub(r'//[^\n]*', '', src)                      # (comments may name the instruction)
        assert '__builtin_amdgcn_global_load_lds' not in code, name
    assert '__builtin_amdgcn_raw_ptr_buffer_load_lds' in open(os.path.join(root, 'sx_flow_kernel.h')).read()


def test_host_code_under_address_and_ub_sanitizers():
    """SURVEY 5, sanitizer row: `make asan` builds the library's HOST code (every launcher and argument validator, --offload-host-only,
    no device code) with -fsanitize=address,undefined and links tests/asan_driver.cpp, which drives every sx_* entry point's argument
    validation without a GPU: plain bad arguments, every single-field mutation of valid cfg 2 / 3 / 4 / backward / wide / MLP programs
    with the out-of-range values the round-3 fuzz found, and 60,000 random programs.  No sanitizer report, no accepted bad call."""
    import subprocess
    csrc = os.path.join(ROOT, 'stribor_amd', 'csrc')
    b = subprocess.run(['make', '-C', csrc, '-j', str(min(8, os.cpu_count() or 1)), 'asan'], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([os.path.join(csrc, 'asan', 'asan_driver')], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-3000:]
    assert ' 0 failures' in r.stdout, r.stdout[-1500:]


def test_error_reporting_modes_and_the_asserting_op_rule():
    """STRIBOR_SYNC_ERRORS / set_sync_errors: 'grad' (default), '1' (every launch), '0' (never); only flows that hold the reference's
    asserting op (the rational-quadratic spline, rational_quadratic_spline.py:175-178,223) synchronise at the end of a training call."""
    from stribor_amd import _hip
    assert _hip._parse_sync_mode('grad') == 'grad' and _hip._parse_sync_mode(' Train ') == 'grad'
    assert _hip._parse_sync_mode('0') == '0' and _hip._parse_sync_mode('') == '0' and _hip._parse_sync_mode(False) == '0'
    assert _hip._parse_sync_mode('1') == '1' and _hip._parse_sync_mode(True) == '1' and _hip._parse_sync_mode('yes') == '1'
    old = _hip.set_sync_errors(True)
    try:
        assert _hip._sync_errors and _hip._sync_mode == '1'
        assert _hip.set_sync_errors('grad') == '1' and not _hip._sync_errors and _hip._sync_mode == 'grad'
        assert _hip.set_sync_errors(False) == 'grad' and _hip._sync_mode == '0'
    finally:
        _hip.set_sync_errors(old)
    dim = 6
    rq = st.NormalizingFlow(st.UnitNormal(dim), [st.Coupling(st.Spline(dim, 4, latent_net=st.net.MLP(dim, [8], dim * 11), lower=-3, upper=3,
                                                                       spline_type='quadratic'), mask='ordered_right_half'), st.Flip()])
    cub = st.NormalizingFlow(st.UnitNormal(dim), [st.Coupling(st.Spline(dim, 4, latent_net=st.net.MLP(dim, [8], dim * 10), lower=-3, upper=3,
                                                                        spline_type='cubic'), mask='ordered_right_half')])
    aff = st.NormalizingFlow(st.UnitNormal(dim), [st.Coupling(st.Affine(dim, latent_net=st.net.MLP(dim, [8], 2 * dim)), mask='ordered_left_half')])
    assert rq._holds_asserting_op() and not cub._holds_asserting_op() and not aff._holds_asserting_op()
    aff.transforms.append(rq.transforms[0])                     # a structure edit re-evaluates the cached answer
    assert aff._holds_asserting_op()


def test_kernel_mode_families_cover_every_mode_once():
    """csrc/sx_flow_types.h: SX_MODE_FAMILY splits the kernel MODEs over three objects per (tiles, hidden tiles, arithmetic); the
    dispatcher (sx_flow_fused.hip) and every launcher (sx_flow_kernel.h: SX_FM) go through the same macro, and the Makefile builds the
    three families."""
    import re
    csrc = os.path.join(ROOT, 'stribor_amd', 'csrc')
    types = open(os.path.join(csrc, 'sx_flow_types.h')).read()
    m = re.search(r'#define SX_MODE_FAMILY\(MODE\) (.*)', types)
    assert m, 'SX_MODE_FAMILY'
    # evaluate the macro itself: a ten-line C program around its definition
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, 'fam.c')
        with open(src, 'w') as f:
            f.write('#include <stdio.h>\n#define SX_MODE_FAMILY(MODE) %s\nint main(void) { for (int m = 0; m <= 20; ++m) printf("%%d ", SX_MODE_FAMILY(m)); return 0; }\n' % m.group(1))
        subprocess.run(['gcc', '-O0', src, '-o', os.path.join(td, 'fam')], check=True)
        values = [int(v) for v in subprocess.run([os.path.join(td, 'fam')], check=True, capture_output=True, text=True).stdout.split()]

    def family(M):
        return values[M]
    kernel = open(os.path.join(csrc, 'sx_flow_kernel.h')).read()
    launched = sorted({int(v) for v in re.findall(r'SX_FM\((\d+)\)', kernel)})
    assert launched == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20], launched
    fam = {M: family(M) for M in launched}
    assert [M for M in launched if fam[M] == 2] == [4, 11]
    # every MODE whose kernel runs spline phases (sx_flow_spline.h / sx_cubic_core.h) is in family 1 -- the only objects whose
    # Makefile prerequisites hold those headers (round 6, ADVICE r5: MODE 10 was in family 0)
    spline_modes = set()
    for line in kernel.splitlines():
        if re.search(r'constexpr bool (RQ|CUB|MIX|RQDEEP)\b.*=', line):
            spline_modes |= {int(v) for v in re.findall(r'MODE == (\d+)', line)}
    assert {3, 10, 12, 13, 18, 19} <= spline_modes, spline_modes
    assert [M for M in launched if fam[M] == 1] == sorted(spline_modes), (fam, spline_modes)
    mk = open(os.path.join(csrc, 'Makefile')).read()
    assert 'FAMS    := 0 1 2' in mk and '-DSX_FAMILY=$(1)' in mk
    fused = open(os.path.join(csrc, 'sx_flow_fused.hip')).read()
    assert 'SX_MODE_FAMILY(mlp_mode)' in fused and '_f2(a)' in fused
