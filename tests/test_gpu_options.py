"""-m gpu: the option table of a context (ft_context_set_option / get_option), and the life-time rule of ft_context_destroy."""
import numpy as np
import pytest

from fasttrack_amd import _capi, orb, synth

pytestmark = pytest.mark.gpu


def test_options_round_trip_and_scope():
    ctx = orb.Context(0)
    table = orb.Context.option_table()
    for name, env, default, _ in table:
        assert ctx.get_option(name) == default or env in __import__("os").environ  # initial value = default unless FT_<NAME> is set
        assert ctx.get_option(env) == ctx.get_option(name)                         # both spellings
    with ctx.options(device_octree=0, pipeline_depth=3):
        assert ctx.get_option("device_octree") == 0 and ctx.get_option("pipeline_depth") == 3
        ex_host = orb.ORBextractor(ctx, 500, 1.2, 8, 20, 7, 320, 240, max_batch=2)
    assert ctx.get_option("device_octree") == 1 and ctx.get_option("pipeline_depth") == 0
    ex_dev = orb.ORBextractor(ctx, 500, 1.2, 8, 20, 7, 320, 240, max_batch=2)
    # an extractor keeps the switches it was created under: same results, different octree
    img = synth.make_image(320, 240, seed=3)
    before = ctx.get_stat("extract.device_octree_batches")[1] if _has(ctx, "extract.device_octree_batches") else 0
    a = ex_host.extract_batch([img, img])
    mid = ctx.get_stat("extract.device_octree_batches")[1] if _has(ctx, "extract.device_octree_batches") else 0
    b = ex_dev.extract_batch([img, img])
    after = ctx.get_stat("extract.device_octree_batches")[1]
    assert mid == before and after == mid + 1
    assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])
    with pytest.raises(_capi.FastTrackError) as e:
        ctx.set_option("no_such_option", 1)
    assert "unknown option" in str(e.value)
    # a second context has its own values
    ctx2 = orb.Context(0)
    ctx.set_option("search_cache", 0)
    assert ctx2.get_option("search_cache") == 2
    ctx.set_option("search_cache", 2)
    ex_host.close()
    ex_dev.close()
    ctx2.close()
    ctx.close()


def _has(ctx, name):
    try:
        ctx.get_stat(name)
        return True
    except Exception:
        return False


def test_context_close_refuses_while_objects_live_and_keeps_everything_valid():
    ctx = orb.Context(0)
    pinned = ctx.pinned_array((16,), np.int32)
    ex = orb.ORBextractor(ctx, 300, 1.2, 4, 20, 7, 160, 120)
    with pytest.raises(_capi.FastTrackError):
        ctx.close()  # the extractor runs on the context's streams
    assert ctx._h is not None
    pinned[:] = 7  # still mapped
    img = synth.make_image(160, 120, seed=1)
    k, d, _ = ex(img)  # and the context still works
    assert len(k) > 0
    ex.close()
    ctx.close()
    assert ctx._h is None
