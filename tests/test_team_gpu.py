"""The persistent per-XCD trajectory kernel (csrc/dit_team.h: one launch per trajectory, one graph per XCD, weights through a register FIFO)
against the launch chain of the same library and, at full size, against the oracle (tests/test_parity_full_size_gpu.py, mode "team").
Here: a two-block denoiser of the reference's width (hidden 1024, 16 heads, 32 nodes) so that the file runs in seconds --
 * one teacher-forced step: decoder logits of the team kernel == the chain's up to the bf16 summation order;
 * the sampled state of that step under injected noise agrees on nearly every entry (ties of the race aside);
 * whole trajectories: well-formed graphs, a seed fixes them, and -- the team path's own property -- a graph's trajectory does not depend on
   the batch it sits in beyond its position (every graph is processed alone by one XCD)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(depth=2, T=10, nodes=32):
    import bench
    args = types.SimpleNamespace(hidden=1024, depth=depth, heads=16, T=T, guide=2.0, nodes=nodes, dtype="bf16")
    return bench.build_model(args, torch.device("cuda"))[0]


def _upper(n_nodes, N):
    i = torch.arange(N)
    return (i.unsqueeze(0) < i.unsqueeze(1)).t().unsqueeze(0) & (i.view(1, N, 1) < n_nodes.view(-1, 1, 1)) & (i.view(1, 1, N) < n_nodes.view(-1, 1, 1))


@pytest.mark.parametrize("B", [1, 3, 8, 11])
def test_team_step_matches_the_launch_chain(B):
    from llamole_amd import synth
    m = _model()
    N, T = 32, 10
    props, text, _ = synth.make_dit_inputs(B, seed=3, max_node=N)
    n_nodes = torch.tensor(([32, 17, 5, 32, 1, 29, 32, 8, 32, 2, 31] * 2)[:B])
    mask = torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)
    um = _upper(n_nodes, N)
    m.begin(props, text, -200.0, n_nodes)
    out = {}
    for mode in (0, 1):
        m.set_option("team", mode)
        m.init_state(*synth.exp_noise(5, T, B, N))
        per = []
        for s in (T - 1, T - 2, 3):
            lx, le = m.denoise_logits(s)
            px, pe = m.step_probs(s)
            m.step(s, *synth.exp_noise(5, s, B, N))
            X, E = m.get_state()
            per.append((lx.cpu(), le.cpu(), px.cpu(), pe.cpu(), X.cpu().long(), E.cpu().long()))
        out[mode] = per
    m.set_option("team", 0)
    for (lx0, le0, px0, pe0, X0, E0), (lx1, le1, px1, pe1, X1, E1) in zip(out[0], out[1]):
        scale = max(float(lx0.abs().max()), float(le0.abs().max()), 1.0)
        ex = float(((lx1 - lx0) * mask.view(1, B, N, 1)).abs().max()) / scale
        ee = float(((le1 - le0) * um.view(1, B, N, N, 1)).abs().max()) / scale
        assert ex <= 2e-2 and ee <= 2e-2, (ex, ee)
        tvx = float((0.5 * (px1 - px0).abs().sum(-1))[mask].max())
        assert tvx <= 5e-2, tvx
        assert torch.equal(E1, E1.transpose(1, 2)) and torch.equal(X1[~mask], X0[~mask])
        assert float((X1 == X0)[mask].float().mean()) >= 0.9
        if int(um.sum()):
            assert float((E1 == E0)[um].float().mean()) >= 0.97
        # the second and third probe start from each mode's own state: states may have drifted apart by then, logits follow


def test_team_trajectory_properties():
    from llamole_amd import synth
    m = _model(depth=2, T=10)
    N = 32
    B = 8
    props, text, _ = synth.make_dit_inputs(B, seed=0, max_node=N)
    n_nodes = torch.tensor([32, 32, 17, 5, 32, 1, 29, 32])

    def run(rows, team, seed=42):
        torch.manual_seed(5)
        m.begin(props[rows], text[rows], -200.0, n_nodes[rows])
        m.set_option("team", team)
        return m.generate_graphs(props[rows], text[rows], -200.0, n_nodes=n_nodes[rows], seed=seed)[0]
    a = run(list(range(B)), 1)
    b = run(list(range(B)), 1)
    assert m.last_run_ms()[1] == 10                      # also checks the kernel's error word (ll_dit_last_run_ms)
    for i, (x, e) in enumerate(a):
        n = int(n_nodes[i])
        assert x.shape == (n,) and e.shape == (n, n) and torch.equal(e, e.t()) and int(torch.diagonal(e).abs().sum()) == 0
        assert int(x.min()) >= 0 and int(x.max()) < 16 and int(e.min()) >= 0 and int(e.max()) < 5
        assert torch.equal(x, b[i][0]) and torch.equal(e, b[i][1])
    other = run(list(range(B)), 1, seed=43)
    assert any(not torch.equal(a[i][1], other[i][1]) for i in range(B))
    # graphs 0..2 alone (batch 3) walk exactly the trajectories they walk inside the batch of 8: same position, same seed, own XCD
    sub = run([0, 1, 2], 1)
    for i in range(3):
        assert torch.equal(sub[i][0], a[i][0]) and torch.equal(sub[i][1], a[i][1])
    # the chain's trajectories are a different rounding of the same model: most entries agree after 10 steps of a 2-block denoiser
    c = run(list(range(B)), 0)
    agree = sum(float((c[i][1] == a[i][1]).float().mean()) for i in range(B) if int(n_nodes[i]) > 1) / sum(int(n) > 1 for n in n_nodes)
    assert agree >= 0.9, agree
    m.set_option("team", 0)


def test_team_kernel_with_sixteen_node_graphs_and_padding_rows():
    """max_nodes = 16: half of every 32-row sequence panel is padding (zero rows staged through out-of-range buffer offsets), F = 96 = six
    output tiles; one teacher-forced step against the chain, then a trajectory."""
    from llamole_amd import synth
    N, T, B = 16, 10, 5
    m = _model(nodes=N)
    props, text, _ = synth.make_dit_inputs(B, seed=4, max_node=N)
    n_nodes = torch.tensor([16, 9, 1, 16, 4])
    mask = torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)
    um = _upper(n_nodes, N)
    m.begin(props, text, -200.0, n_nodes)
    out = {}
    for mode in (0, 1):
        m.set_option("team", mode)
        m.init_state(*synth.exp_noise(6, T, B, N))
        out[mode] = [t.cpu() for t in m.denoise_logits(T - 1)]
    scale = max(float(out[0][0].abs().max()), float(out[0][1].abs().max()), 1.0)
    assert float(((out[1][0] - out[0][0]) * mask.view(1, B, N, 1)).abs().max()) / scale <= 2e-2
    assert float(((out[1][1] - out[0][1]) * um.view(1, B, N, N, 1)).abs().max()) / scale <= 2e-2
    torch.manual_seed(1)
    a = m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=9)[0]
    assert m.last_run_ms()[1] == T
    for i, (x, e) in enumerate(a):
        n = int(n_nodes[i])
        assert x.shape == (n,) and torch.equal(e, e.t()) and int(x.min()) >= 0 and int(x.max()) < 16 and int(e.max()) < 5
    m.set_option("team", 0)


@pytest.mark.parametrize("B,N", [(1, 32), (3, 32), (8, 32), (11, 32), (5, 16), (9, 12)])
def test_proj_ln_team_launch_matches_the_two_launches(B, N):
    """The block's attention projection + AdaLN epilogue as one launch on per-XCD teams (dit_team.h: proj_ln_team_kernel, the default from
    batch 3) against the projection GEMM + ln_mod_res pair of the same library: logits up to the bf16 summation order (full K per tile
    instead of split-K slabs), the sampled state under the same injected noise, and -- launched back to back 3 x depth x 2 times on one
    zero-initialised control block -- that the kernel leaves its counters clean."""
    from llamole_amd import synth
    T = 10
    m = _model(nodes=N)
    props, text, _ = synth.make_dit_inputs(B, seed=3, max_node=N)
    n_nodes = torch.tensor([min(N, v) for v in ([32, 17, 5, 32, 1, 29, 32, 8, 32, 2, 31] * 2)[:B]])
    mask = torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)
    um = _upper(n_nodes, N)
    m.begin(props, text, -200.0, n_nodes)
    out = {}
    for mode in (0, 1):
        m.set_option("proj_ln", mode)
        m.init_state(*synth.exp_noise(5, T, B, N))
        per = []
        for s in (T - 1, T - 2, 3):
            lx, le = m.denoise_logits(s)
            m.step(s, *synth.exp_noise(5, s, B, N))
            X, E = m.get_state()
            per.append((lx.cpu(), le.cpu(), X.cpu().long(), E.cpu().long()))
        out[mode] = per
    m.set_option("proj_ln", 0)
    for (lx0, le0, X0, E0), (lx1, le1, X1, E1) in zip(out[0], out[1]):
        scale = max(float(lx0.abs().max()), float(le0.abs().max()), 1.0)
        ex = float(((lx1 - lx0) * mask.view(1, B, N, 1)).abs().max()) / scale
        ee = float(((le1 - le0) * um.view(1, B, N, N, 1)).abs().max()) / scale
        assert ex <= 2e-2 and ee <= 2e-2, (ex, ee)
        assert torch.equal(E1, E1.transpose(1, 2)) and torch.equal(X1[~mask], X0[~mask])
        assert float((X1 == X0)[mask].float().mean()) >= 0.9
        if int(um.sum()):
            assert float((E1 == E0)[um].float().mean()) >= 0.97


def test_proj_ln_team_trajectories_are_fixed_by_the_seed():
    from llamole_amd import synth
    m = _model(depth=2, T=10)
    N, B = 32, 8
    props, text, _ = synth.make_dit_inputs(B, seed=0, max_node=N)
    n_nodes = torch.tensor([32, 32, 17, 5, 32, 1, 29, 32])

    def run(mode, seed=42):
        torch.manual_seed(5)
        m.set_option("proj_ln", mode)
        return m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=seed)[0]
    a, b, c = run(1), run(1), run(0)
    assert m.last_run_ms()[1] == 10
    # replayed as a hipGraph the kernel meets its control block exactly as the previous launch left it (the last workgroup of every XCC
    # zeroes the counters): the same trajectory as from the launch loop
    torch.manual_seed(5)
    m.set_option("proj_ln", 1)
    g = m.generate_graphs(props, text, -200.0, n_nodes=n_nodes, seed=42, use_graph=True)[0]
    assert m.last_run_ms()[1] == 10
    for i, (x, e) in enumerate(a):
        assert torch.equal(x, g[i][0]) and torch.equal(e, g[i][1])
    for i, (x, e) in enumerate(a):
        assert torch.equal(x, b[i][0]) and torch.equal(e, b[i][1]) and torch.equal(e, e.t())
    agree = sum(float((c[i][1] == a[i][1]).float().mean()) for i in range(B) if int(n_nodes[i]) > 1) / sum(int(n) > 1 for n in n_nodes)
    assert agree >= 0.9, agree
    m.set_option("proj_ln", 0)
