"""legommenders_amd/arena.py on the CPU (the bookkeeping is device-independent): frames close in LIFO order or lazily out of order,
the buffer only grows, growth events stop once the peak has been seen, chunks merge when the arena is empty."""
import torch

from legommenders_amd.arena import Arena


def test_frames_views_growth_and_merge():
    a = Arena("cpu", headroom=1.25)
    f1 = a.push()
    x = a.take(100, 64)                                   # 25 600 B
    y = a.take(3, 5, dtype=torch.int32, zero=True)
    assert x.shape == (100, 64) and x.dtype == torch.float32 and y.dtype == torch.int32 and int(y.abs().sum()) == 0
    x.fill_(1.0)
    y.fill_(7)
    assert float(x.sum()) == 6400.0 and int(y.sum()) == 105        # the two views do not overlap
    base = a.chunks[0].data_ptr()                         # (offsets inside a chunk are 256-byte aligned; hipMalloc aligns the chunk itself)
    assert (x.data_ptr() - base) % 256 == 0 and (y.data_ptr() - base) % 256 == 0
    n0 = a.allocations
    f2 = a.push()
    big = a.take(1 << 16, 16)                             # 4 MB: does not fit the first chunk -> a second one
    assert a.allocations == n0 + 1 and len(a.chunks) == 2
    big.fill_(2.0)
    assert float(x.sum()) == 6400.0                       # earlier views untouched
    f1.release()                                          # out of order: nothing is reclaimed while the frame above is open
    assert len(a.frames) == 2 and a.used > 0
    f2.release()                                          # now both close; the arena is empty and its two chunks merge into one
    assert not a.frames and a.used == 0 and len(a.chunks) == 1
    assert a.chunks[0].numel() >= int(a.peak * 1.25) - 256
    # steady state: the same (or a smaller) request pattern allocates nothing
    n1 = a.allocations
    for _ in range(5):
        f = a.push()
        a.take(100, 64); a.take(1 << 16, 16); a.take(10)
        f.release()
    assert a.allocations == n1 and a.used == 0
    # a frame whose owner dies without releasing it is reclaimed by its finaliser
    f = a.push()
    a.take(1000)
    del f
    import gc
    gc.collect()
    assert not a.frames and a.used == 0
    # reserve(): a caller that knows its capacity sizes the arena ahead of time
    b = Arena("cpu")
    b.reserve(1 << 20)
    n2 = b.allocations
    f = b.push(); b.take(1 << 17); b.take(1 << 16); f.release()
    assert b.allocations == n2


def test_zero_sized_take_and_nested_frames():
    a = Arena("cpu")
    f = a.push()
    e = a.take(0, 8)
    assert e.numel() == 0
    g = a.push()
    t = a.take(16)
    g.release()
    u = a.take(16)                                        # the inner frame's space is handed out again
    assert u.data_ptr() == t.data_ptr()
    f.release()
    assert a.used == 0
