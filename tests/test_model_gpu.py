"""GPU parity of the module stack (SpectralConv, FNO block, SFNO net) against the CPU oracle
with identical weights and inputs.  Everything spectral runs through libmakani_amd.so."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _f64(t):
    t = t.detach().cpu()
    return t.to(torch.complex128) if t.is_complex() else t.double()


def rel(a, b, floor=0.0):
    a, b = _f64(a), _f64(b)
    return (torch.linalg.norm(a - b) / max(torch.linalg.norm(b).item(), floor)).item()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("grids,shapes", [
    (("equiangular", "equiangular"), ((33, 64), (33, 64))),        # same grid: residual is the input
    (("equiangular", "legendre-gauss"), ((33, 64), (16, 32))),     # down-scaling layer: residual = isht(sht(x))
    (("legendre-gauss", "equiangular"), ((16, 32), (33, 64))),     # up-scaling layer
])
def test_spectral_conv_vs_oracle(dev, grids, shapes):
    from makani_amd.sht import RealSHT, InverseRealSHT
    from makani_amd.spectral_convolution import SpectralConv
    from oracle import spectral as osp
    torch.manual_seed(333)
    (ki, ni), (ko, no) = shapes
    L, M, I, O, B = 16, 17, 6, 5, 2
    conv = SpectralConv(RealSHT(ki, ni, L, M, grids[0]), InverseRealSHT(ko, no, L, M, grids[1]), I, O,
                        operator_type="dhconv", bias="constant").to(dev)
    ref = osp.SpectralConv(osp.TorchRealSHT(ki, ni, L, M, grids[0]), osp.TorchInverseRealSHT(ko, no, L, M, grids[1]),
                           I, O, operator_type="dhconv", bias="constant")
    with torch.no_grad():
        ref.bias.normal_()
    conv.load_state_dict(ref.state_dict())
    x = torch.randn(B, I, ki, ni)
    xd = x.to(dev).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    y, r = conv(xd)
    yo, ro = ref(xo)
    assert rel(y, yo) < TOL and rel(r, ro) < TOL
    gy, gr = torch.randn_like(yo), torch.randn_like(ro)
    (y * gy.to(dev)).sum().add((r * gr.to(dev)).sum()).backward()
    (yo * gy).sum().add((ro * gr).sum()).backward()
    assert rel(xd.grad, xo.grad) < TOL
    assert rel(conv.weight.grad, ref.weight.grad) < TOL
    assert rel(conv.bias.grad, ref.bias.grad) < TOL
    # bf16 activations in, bf16 out (spectral_convolution.py:126,146): internals stay fp32
    yb, _ = conv(x.to(dev).to(torch.bfloat16))
    assert yb.dtype == torch.bfloat16
    yob, _ = ref(x.to(torch.bfloat16))
    assert rel(yb.float(), yob.float()) < 1e-2


def test_spectral_conv_production_size_vs_oracle(dev):
    """BASELINE.json configs[1]: one spectral layer on 73 channels at 721x1440 (equiangular) -> 240x480
    (legendre-gauss), 240 x 241 modes, forward + residual against the CPU oracle (tolerance 1e-5)."""
    from makani_amd.sht import RealSHT, InverseRealSHT
    from makani_amd.spectral_convolution import SpectralConv
    from oracle import spectral as osp
    torch.manual_seed(7)
    L, M, C = 240, 241, 73
    conv = SpectralConv(RealSHT(721, 1440, L, M, "equiangular"), InverseRealSHT(240, 480, L, M, "legendre-gauss"), C, C,
                        operator_type="dhconv").to(dev)
    ref = osp.SpectralConv(osp.TorchRealSHT(721, 1440, L, M, "equiangular"),
                           osp.TorchInverseRealSHT(240, 480, L, M, "legendre-gauss"), C, C, operator_type="dhconv")
    conv.load_state_dict(ref.state_dict())
    x = torch.randn(1, C, 721, 1440)
    with torch.no_grad():
        y, r = conv(x.to(dev))
        yo, ro = ref(x)
    assert tuple(y.shape) == (1, C, 240, 480)
    assert rel(y, yo) < TOL and rel(r, ro) < TOL


def test_generic_path_with_foreign_transforms(dev):
    """Duck-typed transforms (RealFFT2 seam, layers.py:219-287) + get_contract_fun('dhconv') dense semantics."""
    import os
    from makani_amd.layers import RealFFT2, InverseRealFFT2
    from makani_amd.spectral_convolution import SpectralConv
    from oracle import spectral as osp
    torch.manual_seed(1)
    f, fi = RealFFT2(16, 32, lmax=10, mmax=9), InverseRealFFT2(16, 32, lmax=10, mmax=9)
    conv = SpectralConv(f, fi, 3, 4, operator_type="dhconv").to(dev)
    x = torch.randn(2, 3, 16, 32)
    y, _ = conv(x.to(dev))
    w = conv.weight.detach().cpu()
    yo = fi(osp.contract_dhconv(f(x), w))
    assert rel(y, yo) < TOL


def test_contraction_seam_vs_reference_golden(dev, golden_dir):
    import os
    from makani_amd.contractions import get_contract_fun
    g = np.load(os.path.join(golden_dir, "ref_contractions.npz"))
    x = torch.from_numpy(g["x"]).to(dev)
    for name in ("dhconv", "diagonal"):
        w = torch.from_numpy(g["w_" + name]).to(dev)
        fn = get_contract_fun(w, implementation="factorized", separable=False, complex=True, operator_type=name)
        y = fn(x, w, separable=False, operator_type=name)
        assert rel(y, torch.from_numpy(g["y_" + name])) < TOL
    with pytest.raises(RuntimeError):   # the reference's separable einsum is ill-formed; same error here
        get_contract_fun(x, implementation="factorized", separable=True, operator_type="dhconv")(x, x[0, :, :, 0])


NET_CASES = [
    dict(inp_shape=(33, 64), out_shape=(33, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2),
    dict(inp_shape=(91, 180), out_shape=(91, 180), scale_factor=3, inp_chans=5, out_chans=5, embed_dim=16, num_layers=3),
]


@pytest.mark.parametrize("kw", NET_CASES)
def test_sfno_net_vs_oracle(dev, kw):
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from oracle import spectral as osp
    torch.manual_seed(333)
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    net.load_state_dict(ref.state_dict(), strict=True)
    B = 2
    x = torch.randn(B, kw["inp_chans"], *kw["inp_shape"])
    tar = torch.randn(B, kw["out_chans"], *kw["out_shape"])
    xd = x.to(dev).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    y, yo = net(xd), ref(xo)
    assert y.shape == yo.shape
    assert rel(y, yo) < TOL
    ((y - tar.to(dev)) ** 2).mean().backward()
    ((yo - tar) ** 2).mean().backward()
    assert rel(xd.grad, xo.grad) < 5 * TOL
    po = dict(ref.named_parameters())
    # some gradients vanish analytically (a bias in front of an instance norm): measure every parameter's
    # error against the typical gradient norm instead of its own ~0 norm
    scale = float(np.median([torch.linalg.norm(_f64(p.grad)).item() for p in po.values()]))
    # a bias in front of an instance norm has an identically zero gradient: the fused path skips that add and hands the
    # parameter an exact zero (every parameter gets a gradient: DDP with find_unused_parameters=False relies on it)
    assert all(p.grad is not None for p in net.parameters())
    errs = {n: rel(p.grad, po[n].grad, floor=1e-1 * scale) for n, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 5 * TOL, (worst, errs[worst])


@pytest.mark.parametrize("extra", [dict(pos_embed="direct"), dict(pos_embed="frequency"), dict(repeat_layers=2),
                                   dict(checkpointing=3), dict(checkpointing=1, pos_embed="frequency", repeat_layers=2),
                                   dict(normalization_layer="layer_norm"), dict(normalization_layer="none")])
def test_sfno_optional_branches_vs_oracle(dev, extra):
    """pos_embed (sfnonet.py:469-501, 606-618), repeat_layers, the activation-checkpointing levels (574-585) and the other
    normalisation choices (371-382: channel-wise layer norm, none): forward, input gradient and every parameter gradient
    against the oracle."""
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from oracle import spectral as osp
    torch.manual_seed(21)
    kw = dict(inp_shape=(33, 64), out_shape=(33, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
    okw = {k: v for k, v in extra.items() if k != "checkpointing"}
    ref = osp.SphericalFourierNeuralOperatorNet(**kw, **okw)
    net = SphericalFourierNeuralOperatorNet(**kw, **extra).to(dev)
    net.load_state_dict(ref.state_dict(), strict=True)
    x, tar = torch.randn(2, 4, 33, 64), torch.randn(2, 3, 33, 64)
    xd, xo = x.to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    y, yo = net(xd), ref(xo)
    assert rel(y, yo) < TOL
    ((y - tar.to(dev)) ** 2).mean().backward()
    ((yo - tar) ** 2).mean().backward()
    assert rel(xd.grad, xo.grad) < 5 * TOL
    po = dict(ref.named_parameters())
    scale = float(np.median([torch.linalg.norm(_f64(p.grad)).item() for p in po.values()]))
    errs = {n: rel(p.grad, po[n].grad, floor=1e-1 * scale) for n, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 5 * TOL, (worst, errs[worst])


def test_sfno_bf16_autocast_runs_and_is_close(dev):
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from oracle import spectral as osp
    torch.manual_seed(5)
    kw = NET_CASES[0]
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    net.load_state_dict(ref.state_dict())
    x = torch.randn(1, 4, 33, 64)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x.to(dev))
    y.float().sum().backward()
    assert rel(y.float(), ref(x)) < 3e-2
    assert all(p.grad is not None for n, p in net.named_parameters())


def test_sfno_bf16_engine_gradients(dev):
    """bf16 autocast step through the pixel-column engine (fused MLP / skip / encoder / decoder nodes): every parameter
    gradient against the fp32 oracle at bf16 accuracy (tolerance 5e-2 of the typical gradient norm)."""
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from oracle import spectral as osp
    torch.manual_seed(9)
    kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=6, out_chans=5, embed_dim=32, num_layers=3)
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    net.load_state_dict(ref.state_dict())
    x, tar = torch.randn(2, 6, 32, 64), torch.randn(2, 5, 32, 64)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x.to(dev))
    ((y.float() - tar.to(dev)) ** 2).mean().backward()
    yo = ref(x)
    ((yo - tar) ** 2).mean().backward()
    assert rel(y.float(), yo) < 3e-2
    po = dict(ref.named_parameters())
    scale = float(np.median([torch.linalg.norm(_f64(p.grad)).item() for p in po.values()]))
    errs = {n: rel(p.grad, po[n].grad, floor=scale) for n, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 5e-2, (worst, errs[worst])


def test_sfno_bf16_fused_mlp_node(dev, monkeypatch):
    """MK_MLP_FUSED=1: the MLP / encoder / decoder of the net on the fused conv -> GELU -> conv kernel (hidden field on chip, forward and
    backward): output and every parameter gradient against the fp32 oracle at bf16 accuracy, and against the unfused engine path."""
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    from oracle import spectral as osp
    torch.manual_seed(9)
    kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=6, out_chans=5, embed_dim=32, num_layers=3)
    ref = osp.SphericalFourierNeuralOperatorNet(**kw)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    net.load_state_dict(ref.state_dict())
    x, tar = torch.randn(2, 6, 32, 64), torch.randn(2, 5, 32, 64)
    yo = ref(x)
    ((yo - tar) ** 2).mean().backward()
    po = dict(ref.named_parameters())
    scale = float(np.median([torch.linalg.norm(_f64(p.grad)).item() for p in po.values()]))
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("MK_MLP_FUSED", fused)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x.to(dev))
        ((y.float() - tar.to(dev)) ** 2).mean().backward()
        res[fused] = (y.detach().float().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
        assert rel(y.float(), yo) < 3e-2
        errs = {n: rel(p.grad, po[n].grad, floor=scale) for n, p in net.named_parameters()}
        worst = max(errs, key=errs.get)
        assert errs[worst] < 5e-2, (fused, worst, errs[worst])
    assert rel(res["1"][0], res["0"][0]) < 2e-2
    for n in res["1"][1]:
        assert rel(res["1"][1][n], res["0"][1][n], floor=scale) < 2e-2, n


def test_engine_arena_is_bit_identical_and_tracks_the_weights(dev, monkeypatch):
    """The per-step arena (every packed weight image in one launch, every weight-gradient buffer in one fill, `ops.EngineArena`)
    against the per-call path (MK_ENGINE_ARENA=0): same bits in the output, the same gradients; weights changed in place
    between two steps -- through an optimizer, and behind autograd's back through `.data` -- are seen by the next step; a
    repeated-weight net (repeat_layers = 2: two uses of one weight per step) still sums its gradients."""
    from makani_amd import ops
    from makani_amd.optim import FusedAdam
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    for extra in (dict(), dict(repeat_layers=2)):
        torch.manual_seed(4)
        kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=6, out_chans=5, embed_dim=32, num_layers=2, **extra)
        net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
        x, tar = torch.randn(2, 6, 32, 64, device=dev), torch.randn(2, 5, 32, 64, device=dev)
        opt = FusedAdam(net.parameters(), lr=1e-2)

        def step():
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
            ((y.float() - tar) ** 2).mean().backward()
            return y.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}

        for round_ in range(3):
            monkeypatch.setenv("MK_ENGINE_ARENA", "1")
            y1, g1 = step()
            assert net.__dict__["_arena"][2].n > 0 and ops._ACTIVE_ARENA is None
            monkeypatch.setenv("MK_ENGINE_ARENA", "0")
            y0, g0 = step()
            assert torch.equal(y1, y0), (extra, round_)
            for n in g1:       # (the weight gradients are sums of fp32 atomics: equal up to the order of the additions)
                assert torch.equal(g1[n], g0[n]) or rel(g1[n], g0[n]) < 1e-5, (extra, round_, n)
            if round_ == 0:
                opt.step()                                  # in-place update through the optimizer
            else:
                with torch.no_grad():
                    net.encoder.fwd[0].weight.data.mul_(1.5)    # ... and behind autograd's back
        del net, opt


def test_hip_graph_capture_replay(dev):
    """The reference's capture sequence, literally (makani/utils/trainer.py:109-148): warm-ups on the capture stream,
    ``static_loss`` of the last warm-up still alive when ``capture_begin()`` runs (it is released inside the capture),
    ``gc.collect(); empty_cache()``, capture forward + loss + backward on that stream, replay; the optimizer stays
    outside.  Every launch goes to the capturing stream, nothing allocates with hipMalloc or synchronises; replays
    reproduce the eager numbers.

    Round-1 note: an eager step on the DEFAULT stream whose loss stays alive across the capture segfaults in
    ``capture_end`` -- with plain torch modules too (tools/capture_diag.py, profiles/r02_capture_diag.txt): the live
    graph keeps AccumulateGrad nodes bound to the legacy default stream, autograd then syncs the capturing stream
    with it.  The reference never does that (all its warm-ups run on the capture stream), nor does this test."""
    import gc
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    torch.manual_seed(7)
    kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    static_inp = torch.zeros(2, 4, 32, 64, device=dev)
    static_tar = torch.zeros(2, 3, 32, 64, device=dev)
    x, tar = torch.randn(2, 4, 32, 64, device=dev), torch.randn(2, 3, 32, 64, device=dev)
    static_inp.copy_(x)
    static_tar.copy_(tar)

    capture_stream = torch.cuda.Stream()
    capture_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(capture_stream):
        for _ in range(3):
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                static_pred = net(static_inp)
                static_loss = ((static_pred.float() - static_tar) ** 2).mean()
            static_loss.backward()
        capture_stream.synchronize()
        ref_loss = static_loss.item()                      # eager numbers of the last warm-up
        ref_grads = {n: p.grad.clone() for n, p in net.named_parameters()}
        gc.collect()
        torch.cuda.empty_cache()
        graph = torch.cuda.CUDAGraph()
        net.zero_grad(set_to_none=True)
        graph.capture_begin()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            static_pred = net(static_inp)
            static_loss = ((static_pred.float() - static_tar) ** 2).mean()
        static_loss.backward()
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(capture_stream)

    for _ in range(2):
        static_inp.copy_(x)          # static input (same values): replay must reproduce the eager numbers
        static_tar.copy_(tar)
        graph.replay()
    torch.cuda.synchronize()
    assert abs(static_loss.item() - ref_loss) <= 1e-6 * abs(ref_loss)
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        assert torch.equal(p.grad, ref_grads[n]) or rel(p.grad, ref_grads[n]) < 1e-5, n
    # new data through the same graph
    x2 = torch.randn_like(x)
    static_inp.copy_(x2)
    graph.replay()
    torch.cuda.synchronize()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        eager = ((net(x2).float() - tar) ** 2).mean()
    assert abs(static_loss.item() - eager.item()) <= 1e-5 * abs(eager.item())


def test_microbatch_runner_matches_single_stream(dev):
    """Two micro-batches on two HIP streams (makani_amd/pipeline.py) give the loss and gradients of the plain step."""
    from makani_amd.pipeline import MicroBatchRunner
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    torch.manual_seed(3)
    kw = dict(inp_shape=(64, 128), out_shape=(64, 128), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=16, num_layers=3)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    x, t = torch.randn(4, 4, 64, 128, device=dev), torch.randn(4, 3, 64, 128, device=dev)

    def loss_of(sl):
        return ((net(x[sl]) - t[sl]) ** 2).sum() / 4

    net.zero_grad(set_to_none=True)
    l_ref = loss_of(slice(0, 4))
    l_ref.backward()
    g_ref = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for _ in range(3):      # repeat: stream hazards are timing dependent
        net.zero_grad(set_to_none=True)
        runner = MicroBatchRunner(2)
        l_mb = runner.forward(lambda j: loss_of(slice(2 * j, 2 * j + 2)))
        l_mb.backward()
        runner.sync()
        torch.cuda.synchronize()
        assert abs(l_mb.item() - l_ref.item()) < 1e-5 * abs(l_ref.item())
        scale = float(np.median([g.norm().item() for g in g_ref.values()]))
        for n, p in net.named_parameters():
            if n in g_ref:
                assert rel(p.grad, g_ref[n], floor=0.1 * scale) < 2e-5, n


# ---------------------------------------------------------------------------------------------------------------------
# Review items of round 2: each fix has its test
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,on_engine", [(768, 1536, False), (512, 2048, False), (1536, 768, False), (768, 768, True)])
def test_conv1x1_wide_layers_train_or_fall_back(dev, cin, cout, on_engine):
    """the engine was chosen from the forward shape alone; the data gradient packs the TRANSPOSED GEMM
    (K' = out_channels) and a 768 -> 1536 layer (`sfno_dhealy_73ch_edim768`, icml_models.yaml:277) raised in backward, a 2048-row
    layer already in forward.  `ops.pce_supported_train` checks both orientations and the four-pass limit: a layer trains
    on the engine only where BOTH GEMMs fit (in and out <= 768) and falls back as a whole where one does not -- forward, input gradient and weight gradient against
    the fp32 convolution."""
    from makani_amd import ops
    from makani_amd.layers import Conv1x1
    assert ops.pce_supported_train(cout, cin) == on_engine
    torch.manual_seed(5)
    conv = Conv1x1(cin, cout, bias=True).to(dev)
    x = torch.randn(1, cin, 24, 40, device=dev, requires_grad=True)
    g = torch.randn(1, cout, 24, 40, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    y.float().backward(g)
    xr = x.detach().clone().requires_grad_(True)
    wr = conv.weight.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, conv.bias.detach())
    gx_ref, gw_ref = torch.autograd.grad(yr, (xr, wr), g)
    assert rel(y, yr) < 1e-2
    assert rel(x.grad, gx_ref) < 1.5e-2
    assert rel(conv.weight.grad, gw_ref) < 1.5e-2


def test_legendre_operand_past_2_gib(dev):
    """the branch-free stagers addressed the k-major operand through ONE buffer descriptor with 32-bit byte
    offsets spanning the whole tensor; past 2 GiB in-range lanes read zeros without an error.  The descriptor is rebased per
    k-step now.  A latitude-major operand of 2.4 GiB: the spectrum at the last channels and modes (whose rows lie past 2^31 bytes
    from the base) against the float64 contraction."""
    from makani_amd import ops
    K, L, M, BC = 96, 48, 49, 65536
    assert K * M * BC * 8 > 2 ** 31
    tab = ops.legendre_table("legendre-gauss", K, L, M, True).to(dev)
    xf = torch.empty(K, M, BC, dtype=torch.complex64, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    torch.view_as_real(xf).normal_(generator=g)
    c = ops.legendre_fwd_raw(xf, tab, L, 0, None, True)              # [L, M, BC]
    for m in (0, 17, 47):
        for ch in (slice(0, 64), slice(BC - 64, BC)):
            want = torch.einsum("lk,kc->lc", tab[m, :, :K].double().to(torch.complex128), xf[:, m, ch].to(torch.complex128))
            got = c[m:, m, ch]
            assert rel(got, want[m:]) < 1e-5, (m, ch)
    # and back: synthesis reads the 2.4 GiB spectrum-side operand the same way
    tabi = ops.legendre_table("legendre-gauss", K, L, M, False).to(dev)
    big = torch.zeros(L, M, BC, dtype=torch.complex64, device=dev)
    big.copy_(c)
    l = torch.arange(L, device=dev).view(-1, 1, 1)
    mm = torch.arange(M, device=dev).view(1, -1, 1)
    big = torch.where(l >= mm, big, torch.zeros((), dtype=big.dtype, device=dev))
    y = ops.legendre_inv_raw(big, tabi, K, 0, None, True)            # [K, M, BC]
    for m in (0, 47):
        ch = slice(BC - 64, BC)
        want = torch.einsum("lk,lc->kc", tabi[m, :, :K].double().to(torch.complex128), big[:, m, ch].to(torch.complex128))
        assert rel(y[:, m, ch], want) < 1e-5, m


def test_hip_graph_capture_after_one_warmup(dev):
    """the first cache hit of a Legendre tile image queried its build event -- illegal inside a capture (and so is
    waiting for it: the event belongs to uncaptured work).  With
    ONE warm-up iteration the first hit after the build lies inside the capture (the three warm-ups of the test above hid it)."""
    import gc
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    torch.manual_seed(7)
    kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=8, num_layers=2)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    x = torch.randn(2, 4, 32, 64, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.no_grad():
            ref = net(x).clone()                             # the one warm-up: builds the images, records their events
        gc.collect()
        graph = torch.cuda.CUDAGraph()
        graph.capture_begin()
        with torch.no_grad():
            out = net(x)
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_fused_adam_state_dict_round_trip_and_foreign_layout(dev):
    """a state dict of another layout (plain torch Adam) must be refused, not ignored; lr changes made on
    ``param_groups`` must reach the large-tensor pass; the own layout restores bit-exactly."""
    from makani_amd.optim import FusedAdam
    torch.manual_seed(0)
    big = torch.nn.Parameter(torch.randn(1 << 21, device=dev))            # the large-tensor path
    small = torch.nn.Parameter(torch.randn(64, device=dev))
    opt = FusedAdam([big, small], lr=1e-3)
    for _ in range(2):
        big.grad, small.grad = torch.randn_like(big), torch.randn_like(small)
        opt.step()
    sd = opt.state_dict()
    with pytest.raises(ValueError):
        opt.load_state_dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))]).state_dict())
    assert opt.param_groups is opt.param_groups                            # ONE persistent list: a scheduler's change sticks
    for gr in opt.param_groups:
        gr["lr"] = 0.0                                                     # ... and reaches both passes
    before = (big.detach().clone(), small.detach().clone())
    big.grad, small.grad = torch.randn_like(big), torch.randn_like(small)
    opt.step()
    assert torch.equal(big, before[0]) and torch.equal(small, before[1])
    opt2 = FusedAdam([big, small], lr=1e-3)
    opt2.load_state_dict(sd)
    for a, b in zip(opt2.state_dict()["big"], sd["big"]):
        assert torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"]) and a["step"] == b["step"]


def test_capture_after_eager_steps_on_the_default_stream(dev):
    """`bench.py --graph`'s sequence: eager warm-up steps on the DEFAULT stream, nothing of their autograd graphs kept, then
    `torch.cuda.graph` (capture on a side stream) of forward + loss + backward, replay.  Round 3 broke it once: the per-step arena
    kept `weight.view(out, in)` tensors, whose grad_fn pins each parameter's AccumulateGrad node to the stream of the first
    step -- `capture_end` then dies (the round-1 crash).  The arena keeps detached views now."""
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    torch.manual_seed(11)
    kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=16, num_layers=2)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    x, tar = torch.randn(2, 4, 32, 64, device=dev), torch.randn(2, 3, 32, 64, device=dev)

    def fwd_bwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = net(x)
        loss = ((pred.float() - tar) ** 2).mean()
        loss.backward()
        return loss

    for _ in range(2):                       # eager, default stream; the loss is dropped at once
        net.zero_grad(set_to_none=True)
        float(fwd_bwd())
    ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = fwd_bwd()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(static_loss).all()
    for n, p in net.named_parameters():
        assert p.grad is not None and rel(p.grad, ref[n], floor=1e-6) < 1e-3, n


def test_fp32_inference_path_matches_training_forward(dev):
    """Grad mode off, fp32: the convolutions carry their bias / GELU in the GEMM epilogue (`mk_conv1x1_x3_bias_act`) and the bf16 MLP
    node keeps no pre-activation -- the output must be the training-mode forward's (fp32 rounding aside)."""
    from makani_amd.sfnonet import SphericalFourierNeuralOperatorNet
    torch.manual_seed(13)
    kw = dict(inp_shape=(32, 64), out_shape=(32, 64), scale_factor=2, inp_chans=4, out_chans=3, embed_dim=16, num_layers=2, bias=False)
    net = SphericalFourierNeuralOperatorNet(**kw).to(dev)
    x = torch.randn(2, 4, 32, 64, device=dev)
    y_train = net(x).detach()
    with torch.no_grad():
        y_eval = net(x)
    assert rel(y_eval, y_train) < 2e-6
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb_train = net(x).detach()
        with torch.no_grad():
            yb_eval = net(x)
    assert torch.equal(yb_eval, yb_train)
