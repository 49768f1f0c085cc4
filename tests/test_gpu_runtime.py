"""The fl_* runtime mini-ABI on the GPU (include/bimocq_gpu.h section 2): allocation is zero-filled, copies round-trip,
events time the compute stream, asynchronous downloads deliver what was on the device when they were requested,
options read back, errors latch and clear."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import gpufluidsimulation_amd as bq
    l = bq.hip_lib()
    assert l.fl_init(0) == 0, l.fl_last_error_string()
    yield l
    bq.check()


def test_malloc_is_zero_filled_and_copies_round_trip(lib):
    n = 1 << 20
    p = lib.fl_malloc(4 * n)
    assert p
    back = np.ones(n, np.float32)
    lib.fl_memcpy_d2h(back.ctypes.data, p, 4 * n)
    assert not back.any()                                  # allocGPUBuffer semantics: cudaMalloc + cudaMemset(0)
    src = np.arange(n, dtype=np.float32)
    lib.fl_memcpy_h2d(p, src.ctypes.data, 4 * n)
    q = lib.fl_malloc(4 * n)
    lib.fl_memcpy_d2d(q, p, 4 * n)
    lib.fl_memset(p, 0, 4 * n)
    lib.fl_memcpy_d2h(back.ctypes.data, q, 4 * n)
    assert np.array_equal(back, src)
    lib.fl_memcpy_d2h(back.ctypes.data, p, 4 * n)
    assert not back.any()
    lib.fl_free(p); lib.fl_free(q)


def test_events_time_the_compute_stream(lib):
    n = 1 << 26
    p = lib.fl_malloc(4 * n)
    e0, e1 = lib.fl_event_create(), lib.fl_event_create()
    lib.fl_event_record(e0)
    for _ in range(20):
        lib.fl_memset(p, 0, 4 * n)
    lib.fl_event_record(e1)
    ms = lib.fl_event_elapsed_ms(e0, e1)
    assert 0.05 < ms < 200.0, ms                           # 20 x 256 MiB of fills: tens of microseconds each at least
    lib.fl_event_destroy(e0); lib.fl_event_destroy(e1)
    lib.fl_free(p)


def test_async_download_snapshots_the_request_point(lib):
    n = 1 << 22
    dev = lib.fl_malloc(4 * n)
    host = lib.fl_malloc_host(4 * n)
    assert dev and host
    a = np.full(n, 3.0, np.float32)
    lib.fl_memcpy_h2d(dev, a.ctypes.data, 4 * n)
    ticket = lib.fl_download_begin(host, dev, 4 * n)
    assert ticket
    # the copy stream is ordered after the compute work queued so far; compute work queued LATER does not wait for the
    # download, so the caller must keep the source intact until the ticket is waited on -- the solver's dump path
    # downloads a device snapshot of Density for that reason (the next advance() rewrites Density in place:
    # tests/test_gpu_solver.py::test_async_dump_survives_immediate_overwrite)
    assert lib.fl_download_wait(ticket) == 0
    got = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_float)), shape=(n,))
    assert np.array_equal(got, a)
    assert lib.fl_download_wait(None) != 0                 # a null ticket is an error code, not a crash
    lib.fl_free_host(host); lib.fl_free(dev)


def test_options_read_back_and_errors_latch(lib):
    import gpufluidsimulation_amd as bq
    L = bq._lib
    defaults = {L.FL_OPT_RESIDUAL_STRIDE: 0, L.FL_OPT_SKIP_UNIT_BLEND: 1, L.FL_OPT_JACOBI_VARIANT: 0, L.FL_OPT_STRUCTURED_MAPS: 1,
                L.FL_OPT_JACOBI_FUSE: 1, L.FL_OPT_MGCG_GRAPH: 1, L.FL_OPT_FAST_LERP: 0}
    for opt, want in defaults.items():
        assert lib.fl_get_option(opt) == want, opt
    lib.fl_set_option(L.FL_OPT_JACOBI_VARIANT, 3)
    assert lib.fl_get_option(L.FL_OPT_JACOBI_VARIANT) == 3
    lib.fl_set_option(L.FL_OPT_JACOBI_VARIANT, 0)
    assert lib.fl_get_option(12345) == -1
    lib.fl_clear_error()
    lib.fl_memset(None, 0, 16)                             # bad argument: latched, not thrown
    assert lib.fl_last_error() == L.FL_ERR_BAD_ARGUMENT and lib.fl_last_error_string()
    lib.fl_memset(None, 0, 16)
    lib.fl_clear_error()
    assert lib.fl_last_error() == 0
