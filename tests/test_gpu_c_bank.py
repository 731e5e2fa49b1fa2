"""bank maintenance kernels (append / gather / export / rank / eviction / consolidation pieces)."""
import pytest
import torch

import emu_ops
from deva.hip import ops
from gpu_util import dev, max_err, rand, to_dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize('c,n', [(64, 48), (512, 1620), (1, 37), (64, 33), (7, 1)])
def test_append_export_roundtrip(c, n):
    g = torch.Generator().manual_seed(1)
    src = rand(g, c, n)
    arena = torch.full((n + 50, c), -7.0, device=dev())
    ops.bank_append(to_dev(src), arena, 13)
    torch.cuda.synchronize()
    assert torch.equal(arena[13:13 + n].cpu(), src.t())
    assert (arena[:13] == -7).all() and (arena[13 + n:] == -7).all()
    back = ops.bank_export(arena[13:13 + n], n)
    assert torch.equal(back.cpu(), src)


def test_gather_rows():
    g = torch.Generator().manual_seed(2)
    src = rand(g, 200, 64)
    rows = torch.randperm(200, generator=g)[:77].int()
    dst = torch.zeros(77, 64, device=dev())
    ops.bank_gather_rows(to_dev(src), to_dev(rows), dst, 77)
    assert torch.equal(dst.cpu(), src[rows.long()])
    dst1 = torch.zeros(50, device=dev())
    v = rand(g, 300)
    ops.bank_gather_rows(to_dev(v)[100:150], None, dst1, 50)
    assert torch.equal(dst1.cpu(), v[100:150])


@pytest.mark.parametrize('n', [1, 5, 300, 1000, 8100])
@pytest.mark.parametrize('desc', [False, True])
def test_rank_is_stable_sort_position(n, desc):
    g = torch.Generator().manual_seed(3 + n)
    x = (torch.rand(n, generator=g) * 20).round() / 20  # many exact ties
    want, _ = emu_ops.rank(x, n, desc)
    got, _ = ops.rank(to_dev(x), n, desc)
    assert torch.equal(got.cpu(), want)
    use, life = torch.rand(n, generator=g), torch.rand(n, generator=g) + 0.5
    want, wx = emu_ops.rank(use, n, desc, life=life)
    got, gx = ops.rank(to_dev(use), n, desc, life=to_dev(life))
    assert torch.equal(gx.cpu(), wx) and torch.equal(got.cpu(), want)
    if n >= 5:
        k = min(128, n)
        sel = ops.rank_select(got, k)
        assert torch.equal(sel.cpu(), emu_ops.rank_select(want, k))


@pytest.mark.parametrize('n,n_remove', [(64, 16), (3000, 1), (3000, 2999), (5000, 1234)])
def test_evict_select(n, n_remove):
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(n, generator=g) * 50).round() / 50
    r, _ = emu_ops.rank(x, n, False)
    widx, wcount = emu_ops.evict_select(x, r, n_remove)
    gidx, gcount = ops.evict_select(to_dev(x), to_dev(r), n_remove)
    c = int(gcount.item())
    assert c == int(wcount.item())
    assert torch.equal(gidx[:c].cpu(), widx[:c])


@pytest.mark.parametrize('nc,p', [(240, 32), (1000, 128), (77, 5)])
def test_similarity_dense_and_softmax_columns(nc, p):
    g = torch.Generator().manual_seed(5)
    key, shr, sel = rand(g, nc, 64, scale=2.0), torch.rand(nc, generator=g) + 1, torch.rand(nc, 64, generator=g)
    proto = torch.randperm(nc, generator=g)[:p].int()
    want = emu_ops.similarity_dense(key, shr, sel, proto, nc)
    got = ops.similarity_dense(to_dev(key), to_dev(shr), to_dev(sel), to_dev(proto), nc)
    err = max_err(got, want)
    print(f'similarity_dense {nc}x{p}: max abs err {err:.3e} (|ref| max {want.abs().max():.3e})')
    assert err <= 1e-5 * want.abs().max().item()
    assert (got[:, p:] == 0).all()
    want = emu_ops.softmax_columns(want.clone(), p)
    got = ops.softmax_columns(got, p)
    assert max_err(got, want) <= 1e-6
    # prototype readout as a GEMM through deva_conv2d
    vals = rand(g, nc, 512)
    pc = ops.PackedConv(got, None, nc, p, got.shape[1], 1, 1)
    out = ops.conv2d(pc, to_dev(vals).reshape(1, nc, 1, 512)).view(p, 512)
    ref = want[:, :p].t() @ vals
    assert max_err(out, ref) <= 1e-5 * max(1.0, ref.abs().max().item())


# (240, 320, 150, 140): 151 x 141 label pairs = 85 KB, beyond the LDS histogram -> global-atomic kernel
@pytest.mark.parametrize('h,w,n_our,n_new', [(64, 80, 3, 4), (480, 864, 8, 12), (17, 5, 0, 2), (33, 47, 5, 0),
                                             (240, 320, 150, 140)])
def test_label_histogram_and_merge_paint(h, w, n_our, n_new):
    g = torch.Generator().manual_seed(h + n_our)
    ours = torch.randint(0, n_our + 1, (h, w), generator=g)
    new_ids = (torch.randperm(5000, generator=g)[:n_new] + 300).long()
    pick = torch.randint(0, n_new + 2, (h, w), generator=g)  # n_new -> background 0, n_new+1 -> unlisted id
    table = torch.cat([new_ids, torch.tensor([0, 77777])])
    news = table[pick]
    want = emu_ops.label_histogram(ours, news, new_ids, n_our)
    got = ops.label_histogram(to_dev(ours), to_dev(news), to_dev(new_ids), n_our)
    assert torch.equal(got.cpu(), want)
    assert int(want.sum()) == h * w
    our_order = torch.randint(-1, 6, (n_our + 1,), generator=g).int()
    our_label = torch.randint(1, 400, (n_our + 1,), generator=g)
    new_order = torch.randint(-1, 6, (n_new,), generator=g).int()
    new_label = torch.randint(1, 400, (n_new,), generator=g)
    out_ids = torch.unique(torch.cat([our_label, new_label]))[:6]
    want = emu_ops.merge_paint(ours, news, new_ids, our_order, our_label, new_order, new_label, out_ids)
    got = ops.merge_paint(to_dev(ours), to_dev(news), to_dev(new_ids), to_dev(our_order), to_dev(our_label),
                          to_dev(new_order), to_dev(new_label), to_dev(out_ids))
    assert got.shape == want.shape and torch.equal(got.cpu(), want)


def test_lut_remap_and_tmp_to_obj_cls():
    from deva.inference.object_manager import ObjectManager
    g = torch.Generator().manual_seed(5)
    om = ObjectManager()
    om.add_new_objects([7, 3, 250])
    mask = torch.randint(0, 4, (37, 53), generator=g)
    want = om.tmp_to_obj_cls(mask)               # host path
    got = om.tmp_to_obj_cls(to_dev(mask))        # device path (deva_lut_remap)
    assert torch.equal(got.cpu(), want)
    assert set(want.unique().tolist()) <= {0, 7, 3, 250}


@pytest.mark.parametrize('c,h,w,size', [(3, 40, 56, None), (6, 480, 864, (1080, 1920)), (2, 33, 47, (97, 61)),
                                        (4, 96, 128, (48, 64)), (1, 8, 8, (20, 20))])
def test_index_mask_resize_argmax_lut(c, h, w, size):
    """fused output tail vs F.interpolate(bilinear, align_corners=False) -> argmax -> table on the CPU;
    a differing label is tolerated only where the two best resized probabilities tie to 1e-6"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(c * 1000 + h)
    prob = torch.softmax(torch.randn(c, h, w, generator=g) * 2, dim=0)
    prob[:, :3, :5] = 1.0 / c  # exact ties: the first maximum must win
    lut = torch.tensor([0] + [100 + 7 * i for i in range(1, c)], dtype=torch.int64)
    got = ops.index_mask(to_dev(prob), size, to_dev(lut)).cpu()
    res = prob if size is None else F.interpolate(prob.unsqueeze(1), size, mode='bilinear', align_corners=False)[:, 0]
    want = lut[torch.argmax(res, dim=0)]
    assert got.shape == want.shape and got.dtype == torch.int64
    diff = got != want
    if diff.any():
        top2 = res.topk(min(2, c), dim=0)[0]
        margin = (top2[0] - top2[-1])[diff]
        assert margin.max().item() <= 1e-6, f'{int(diff.sum())} labels differ with a decisive margin'
    if size is None:  # no table: plain argmax, ties -> channel 0
        assert int(ops.index_mask(to_dev(prob)).cpu()[:3, :5].abs().sum()) == 0


def test_usage_init():
    use = torch.full((5000,), 3.0, device=dev())
    life = torch.full((5000,), 4.0, device=dev())
    ops.usage_init(use[100:4100], life[100:4100])
    torch.cuda.synchronize()
    want_u, want_l = torch.full((5000,), 3.0), torch.full((5000,), 4.0)
    want_u[100:4100] = 0.0
    want_l[100:4100] = 1e-7
    assert torch.equal(use.cpu(), want_u) and torch.equal(life.cpu(), want_l)

