"""The drop-in boundary without a GPU: libdeva_hip.so loads, exports every entry point that
include/deva_hip.h declares (and nothing else under the `deva_` prefix), the ctypes binding lists
exactly those, the struct mirror has the C layout, and argument validation fails loudly before any
launch.  No compute is called here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'deva_hip.h')


def _declared():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    text = re.sub(r'//[^\n]*', '', text)
    return sorted(set(re.findall(r'\b(deva_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def library():
    from deva import hip
    if not os.path.exists(hip.LIB_PATH):  # the driver builds first; standalone runs build here
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(hip.LIB_PATH)


def test_header_symbols_are_exported(library):
    names = _declared()
    assert len(names) >= 33
    for n in names:
        assert hasattr(library, n), f'{n} declared in include/deva_hip.h but not exported'


def test_no_undeclared_entry_points(library):
    from deva import hip
    out = subprocess.run(['nm', '-D', '--defined-only', hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r'\bT (deva_[a-z0-9_]+)$', out, flags=re.M)))
    assert exported == _declared()


def test_binding_covers_the_header():
    from deva import hip
    assert sorted(hip.SIGNATURES) == _declared()


def test_conv_desc_mirror_has_the_c_layout(tmp_path):
    """compile a probe that prints sizeof/offsetof of struct deva_conv_desc and compare with ctypes"""
    from deva.hip import ConvDesc
    fields = [f[0] for f in ConvDesc._fields_]
    src = tmp_path / 'probe.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "deva_hip.h"\nint main(void){\n'
                   'printf("%zu\\n", sizeof(deva_conv_desc));\n' +
                   ''.join(f'printf("%zu\\n", offsetof(deva_conv_desc, {f}));\n' for f in fields) + 'return 0;}\n')
    exe = tmp_path / 'probe'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(ConvDesc)
    assert vals[1:] == [getattr(ConvDesc, f).offset for f in fields]


def test_version_and_argument_errors_without_a_gpu(library):
    from deva import hip
    L = hip.lib()
    assert L.deva_hip_version() == hip.ABI_VERSION
    assert L.deva_conv2d(None, None) != 0
    assert b'deva_conv2d' in L.deva_hip_last_error()
    assert L.deva_affinity_topk(None, None, 0, None, None, 0, None, None, 0, 30, 1, None, None) != 0
    assert b'deva_affinity_topk' in L.deva_hip_last_error()
    # pure host-side helpers
    assert L.deva_affinity_workspace(1620, 30, 4) == 4 * 1620 * 64 + 4 * 1620 // 2  # keys + 32-bit list lengths
    assert 1 <= L.deva_affinity_default_splits(10000, 8160) <= 32
    assert L.deva_affinity_default_splits(1620, 1620) == 3  # small frame, short bank: one 8-wave workgroup per CU
    assert L.deva_affinity_force_shape(2) == 0
    assert L.deva_affinity_default_splits(1620, 1620) == 26  # per-wave lists: <= 64 tokens per range, no filtering needed
    assert L.deva_affinity_force_shape(9) != 0 and L.deva_affinity_force_shape(0) == 0


def test_read_policy_and_scratch_sizes_are_host_side():
    """host-only entry points of the read path (no GPU work): the pre-filter policy and the scratch layout"""
    from deva.hip import lib
    L = lib()
    assert L.deva_affinity_prefilter_enabled(2048, 1620, 30) == 0      # small bank: the fp32 kernels
    assert L.deva_affinity_prefilter_enabled(10000, 300, 30) == 0      # few scores: the fp32 kernels
    assert L.deva_affinity_prefilter_enabled(10000, 8160, 30) == 1
    assert L.deva_affinity_prefilter_enabled(244400, 32400, 30) == 1
    assert L.deva_affinity_force_prefilter(0) == 0 and L.deva_affinity_prefilter_enabled(10000, 8160, 30) == 0
    assert L.deva_affinity_force_prefilter(1) == 0 and L.deva_affinity_force_prefilter(2) != 0
    small, big = L.deva_affinity_read_scratch(10000, 8160, 30), L.deva_affinity_read_scratch(244400, 32400, 30)
    # the scratch always holds the fp32 kernels' hand-over (the fall-back writes there), plus the pre-filter's operands
    assert small > L.deva_affinity_workspace(8160, 30, L.deva_affinity_default_splits(10000, 8160))
    assert big > small and big * 8 < 2 << 30                           # < 2 GiB at the 4K bench shape
    assert L.deva_affinity_force_shape(3) != 0 and L.deva_affinity_force_shape(4) == 0 and L.deva_affinity_force_shape(0) == 0


def test_conv_pack_matches_python_packing(library):
    """deva_conv_pack (host function of the C ABI) and ops.pack_conv produce the same layout: tap-major / 32-channel
    slabs, k-quad interleaved for more than one output channel, K padded to a multiple of 4"""
    import torch
    from deva import hip
    from deva.hip import ops
    L = hip.lib()
    g = torch.Generator().manual_seed(3)
    for cout, cin, k in ((64, 3, 7), (48, 64, 3), (1, 32, 3), (40, 33, 1), (128, 64, 1), (2, 2, 7)):
        w = torch.randn(cout, cin, k, k, generator=g)
        pc = ops.pack_conv(w)
        lay, cpad = ctypes.c_int(-1), ctypes.c_int(-1)
        n = L.deva_conv_pack(w.contiguous().data_ptr(), None, cout, cin, k, k, 1, ctypes.byref(lay), ctypes.byref(cpad))
        assert n == pc.weight.numel() and lay.value == pc.k_layout and cpad.value == pc.cout_pad, (cout, cin, k)
        out = torch.full((n,), float('nan'))
        assert L.deva_conv_pack(w.contiguous().data_ptr(), out.data_ptr(), cout, cin, k, k, 1, ctypes.byref(lay),
                                ctypes.byref(cpad)) == n
        assert torch.equal(out.view_as(pc.weight), pc.weight), (cout, cin, k)
        assert bool(pc.k_layout & hip.KLAYOUT_Q4) == (cout > 1)


def test_conv_pack_f16_matches_python_packing(library):
    """deva_conv_pack_f16 (host function, the --amp weights) and ops.pack_f16 produce the same bytes: fp16 octets
    Wh[K/8][cout_pad][8], 64-channel slabs for 3x3, round-to-nearest-even; layers with cin % 64 != 0 are refused by both"""
    import torch
    from deva import hip
    from deva.hip import ops
    L = hip.lib()
    g = torch.Generator().manual_seed(5)
    for cout, cin, k in ((64, 64, 1), (72, 128, 3), (1536, 64, 3), (40, 192, 1)):
        w = torch.randn(cout, cin, k, k, generator=g) * 3.0
        w.view(-1)[::7] *= 1e-6   # some values in the fp16 subnormal range, some ties of the rounding
        ref = ops.pack_f16(w)
        cpad = ctypes.c_int(-1)
        n = L.deva_conv_pack_f16(w.contiguous().data_ptr(), None, cout, cin, k, k, ctypes.byref(cpad))
        assert n == ref.numel() and cpad.value == (cout + 31) // 32 * 32, (cout, cin, k)
        out = torch.zeros(n, dtype=torch.int16)
        assert L.deva_conv_pack_f16(w.contiguous().data_ptr(), out.data_ptr(), cout, cin, k, k, ctypes.byref(cpad)) == n
        assert torch.equal(out, ref.view(torch.int16)), (cout, cin, k)
    w = torch.randn(64, 96, 3, 3, generator=g)  # 96 % 64 != 0: not an amp layer
    assert ops.pack_f16(w) is None
    assert L.deva_conv_pack_f16(w.contiguous().data_ptr(), None, 64, 96, 3, 3, ctypes.byref(ctypes.c_int(0))) == -1


def test_conv_pack_split_matches_python_packing(library):
    """deva_conv_pack_split (host function, the --f16_split weights) and ops.pack_split produce the same bytes and the same
    scale: hi / lo fp16 planes W[K/8][2][cout_pad][8] of w * 2^e, 32-channel slabs for 3x3, round-to-nearest-even; hi + lo
    reproduces w * 2^e to 2^-22 relative or 2^-25 absolute; layers with cin % 32 != 0 are refused by both"""
    import torch
    from deva import hip
    from deva.hip import ops
    L = hip.lib()
    g = torch.Generator().manual_seed(9)
    for cout, cin, k, gain in ((64, 64, 1, 0.05), (72, 96, 3, 3.0), (1536, 32, 3, 1e-3), (40, 160, 1, 200.0), (64, 513, 1, 0.1)):
        w = torch.randn(cout, cin, k, k, generator=g) * gain
        w.view(-1)[::7] *= 1e-6   # lo planes in the fp16 subnormal range, some exact zeros
        ref, e_ref = ops.pack_split(w)
        cpad, e = ctypes.c_int(-1), ctypes.c_int(-999)
        n = L.deva_conv_pack_split(w.contiguous().data_ptr(), None, cout, cin, k, k, ctypes.byref(cpad), ctypes.byref(e))
        assert n == ref.numel() and cpad.value == (cout + 31) // 32 * 32 and e.value == e_ref, (cout, cin, k)
        assert 2.0**13 <= float(w.abs().max()) * 2.0**e_ref < 2.0**14
        out = torch.zeros(n, dtype=torch.int16)
        assert L.deva_conv_pack_split(w.contiguous().data_ptr(), out.data_ptr(), cout, cin, k, k, ctypes.byref(cpad),
                                      ctypes.byref(e)) == n
        assert torch.equal(out, ref.view(torch.int16)), (cout, cin, k)
        # hi + lo against the scaled weights (layout undone)
        planes = ref.view(-1, 2, cpad.value, 8).float()
        rec = (planes[:, 0] + planes[:, 1]).permute(0, 2, 1).reshape(-1, cpad.value)[:, :cout]  # [K][cout]
        taps = k * k
        wk = (w.reshape(cout, cin // 32, 32, taps).permute(1, 3, 2, 0).reshape(-1, cout) if taps > 1
              else w.reshape(cout, cin).t()) * 2.0**e_ref
        assert rec.shape[0] == (taps * cin + 31) // 32 * 32 and bool((rec[taps * cin:] == 0).all())  # (1x1: zero rows up to a K step)
        rec = rec[:taps * cin]
        assert bool(((rec - wk).abs() <= torch.maximum(wk.abs() * 2.0**-22, torch.tensor(2.0**-25))).all())
    w = torch.randn(64, 48, 3, 3, generator=g)  # 48 % 32 != 0: not a split layer
    assert ops.pack_split(w) == (None, 0)
    assert L.deva_conv_pack_split(w.contiguous().data_ptr(), None, 64, 48, 3, 3, ctypes.byref(ctypes.c_int(0)),
                                  ctypes.byref(ctypes.c_int(0))) == -1


def test_conv_descriptor_validation_of_the_f16_paths(library):
    """deva_conv2d refuses, before any launch: amp outside {0, 1, 2}; the split path (amp = 2 with weights) without a flag;
    k-quad interleaved weights for a single output channel"""
    from deva import hip
    L = hip.lib()
    d = hip.ConvDesc()
    d.in0, d.weight, d.out = 4096, 8192, 12288               # never dereferenced: validation fails first
    d.c0, d.c1, d.batch, d.height, d.width = 32, 0, 1, 8, 8
    d.cout, d.cout_pad, d.k_layout = 64, 64, hip.KLAYOUT_TAP_MAJOR | hip.KLAYOUT_Q4
    d.kh = d.kw = d.stride = 1
    d.amp = 3
    assert L.deva_conv2d(ctypes.byref(d), None) != 0 and b'amp must be' in L.deva_hip_last_error()
    d.amp, d.weight_f16, d.split_flag = 2, 16384, None
    assert L.deva_conv2d(ctypes.byref(d), None) != 0 and b'split_flag' in L.deva_hip_last_error()
    d.amp, d.weight_f16, d.cout, d.cout_pad = 0, None, 1, 32
    assert L.deva_conv2d(ctypes.byref(d), None) != 0 and b'cout > 1' in L.deva_hip_last_error()
    assert L.deva_affinity_bank_prep_bytes(0) == 512
    assert L.deva_affinity_bank_prep_bytes(10000) == 512 + (313 * 9216 + 255) // 256 * 256   # 313 tiles of 32 tokens
