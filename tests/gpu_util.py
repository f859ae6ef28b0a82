"""Helpers for the -m gpu parity tests: thin ctypes drivers over the C ABI (include/difashion_hip.h)
plus torch fp32 references.  References are evaluated on the SAME bf16-rounded operands the kernels
see, so tolerances only have to cover accumulation order and the final bf16 rounding."""
import ctypes as C

import torch

from difashion_amd import _lib

DEV = "cuda"


def release_cached_gpu_memory():
    """Before a test starts CHILD processes that use the GPU: hand the caching allocator's free blocks back to the driver.  After the
    full-size training tests the pytest process sits on > 100 GB of cached HBM; a child that then allocates makes the kernel driver
    evict / migrate that memory (one test took 243 s of which the two children ran 6 s each: gpurun_out/r06_run2, round 6)."""
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0, device=DEV):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device)


def zero_page():
    return torch.zeros(256, dtype=torch.uint8, device=DEV)


def stream():
    return _lib.stream_ptr()


def gemm_desc(*, M, N, W, ldw, a0=None, a0_c=0, a1=None, a1_c=0, conv_src=None, conv_c=0, batch=0, Hin=0, Win=0, stride=1,
              upsample=0, bias=None, rowvec=None, rv_ld=0, rv_off=0, rows_per_b=0, resid=None, act=0, out_mode=0,
              out=None, ld_out=None, force_tile=0, force_split=0, force_order=-1, w_img_stride=0):
    """A filled dfh_gemm_desc; the tensors it points at stay alive on the descriptor (``keep_*``), the output is ``d.keep_out``."""
    d = _lib.GemmDesc()
    if conv_src is not None:
        d.conv_src, d.conv_c, d.conv = conv_src.data_ptr(), conv_c, 1
        d.batch, d.Hin, d.Win, d.stride, d.upsample = batch, Hin, Win, stride, upsample
    if a0 is not None:
        d.a0, d.a0_c = a0.data_ptr(), a0_c
    if a1 is not None:
        d.a1, d.a1_c = a1.data_ptr(), a1_c
    d.W, d.ldw, d.M, d.N = W.data_ptr(), ldw, M, N
    if bias is not None:
        d.bias = bias.data_ptr()
    if rowvec is not None:
        d.rowvec, d.rv_ld, d.rv_off = rowvec.data_ptr(), rv_ld, rv_off
    d.rows_per_b = rows_per_b
    if resid is not None:
        d.resid, d.ld_res = resid.data_ptr(), N
    d.act, d.out_mode = act, out_mode
    if out is None:
        n_out = N // 2 if act == 4 else N
        out = torch.empty((M, n_out), dtype=torch.float32 if out_mode in (2, 3) else torch.bfloat16, device=DEV)
        ld_out = n_out
    d.out, d.ld_out = out.data_ptr(), ld_out
    z = zero_page()
    d.zero_page = z.data_ptr()
    d.force_tile, d.force_split, d.force_order = force_tile, force_split, force_order
    d.w_img_stride = w_img_stride
    need = _lib.raw().dfh_gemm_partial_floats(C.byref(d))
    part = None
    if need:
        part = torch.empty(need, dtype=torch.float32, device=DEV)
        d.partial, d.partial_floats = part.data_ptr(), need
    d.keep_out, d.keep_alive = out, (z, part, W, a0, a1, conv_src, bias, rowvec, resid)
    return d


def gemm(*, gstat=None, gstat_cpg=0, gstat_hw=0, **kw):
    """gstat: a float32 device tensor for the output's GroupNorm statistics (dfh_gemm_gstat); the call then returns (out, rows) with rows = the
    pixel rows per statistics chunk (256 / 128) or 0 when the launch could not write them."""
    d = gemm_desc(**kw)
    out = d.keep_out
    if gstat is not None:
        d.gstat, d.gstat_cpg, d.gstat_hw = gstat.data_ptr(), gstat_cpg, gstat_hw
        written = C.c_int(0)
        _lib.call("dfh_gemm_gstat", C.byref(d), stream(), C.byref(written))
        torch.cuda.synchronize()
        return out, int(written.value)
    _lib.call("dfh_gemm", C.byref(d), stream())
    torch.cuda.synchronize()
    return out


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def nchw(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2).contiguous()


def pack_conv(w_oihw):
    """fp32 OIHW -> bf16 [O][9*I] via the library's packer."""
    O, I = w_oihw.shape[:2]
    out = torch.empty((O, 9 * I), dtype=torch.bfloat16, device=DEV)
    _lib.call("dfh_pack_conv3x3", _lib.ptr(w_oihw.contiguous()), _lib.ptr(out), O, I, 9 * I, 0, stream())
    return out


def rel_err(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.double() - b.double()).abs().max())


def assert_close_bf16(got, ref, what="", rel=6e-3, max_rel=3e-2):
    """bf16 output vs fp32 reference: rel L2 <= rel, and max |err| <= max_rel * max|ref|."""
    r = rel_err(got, ref)
    m = max_err(got, ref) / (float(ref.abs().max()) + 1e-30)
    assert r <= rel and m <= max_rel, f"{what}: rel_l2={r:.3e} (<= {rel}), max={m:.3e} (<= {max_rel})"
