"""Camera::render_progressive of the front end (include/rtmi.hpp): the pass-by-pass preview that stands in for
render_live / render_interactive's pixel callbacks (camera.rs:162-265).  Pass k is the frame with at most caps[k] samples
per pixel, the last pass is Camera::render's frame, every pass is closer to it than the one before, and a callback that
returns False ends the preview."""
import numpy as np
import pytest

from raytracing_amd import _front as rt
from raytracing_amd import scenes

pytestmark = pytest.mark.gpu


def test_passes_refine_towards_the_final_frame():
    world, cam = scenes.chapter11a(240, 120)
    scene = rt.DeviceScene(world)
    final = cam.render(scene, seed=11).numpy()
    seen = []

    def on_pass(k, n, canvas):
        seen.append((k, n, canvas.numpy().copy()))
        return True

    out = cam.render_progressive(scene, on_pass, caps=[1, 4, 16], seed=11).numpy()
    assert [k for k, _, _ in seen] == [0, 1, 2, 3] and all(n == 4 for _, n, _ in seen)
    assert np.array_equal(seen[-1][2], final) and np.array_equal(out, final)
    # pass 0 is the one-sample frame, pass 1 the four-sample frame
    assert np.array_equal(seen[0][2], cam.render(scene, seed=11, max_samples=1).numpy())
    assert np.array_equal(seen[1][2], cam.render(scene, seed=11, max_samples=4).numpy())
    err = [float(np.abs(img - final).mean()) for _, _, img in seen]
    assert err[0] > err[1] > err[2] > err[3] == 0.0


def test_callback_can_stop_the_preview():
    world, cam = scenes.chapter11a(160, 80)
    scene = rt.DeviceScene(world)
    calls = []
    out = cam.render_progressive(scene, lambda k, n, cv: calls.append(k) or k < 1, caps=[1, 4, 16], seed=5).numpy()
    assert calls == [0, 1]
    assert np.array_equal(out, cam.render(scene, seed=5, max_samples=4).numpy())


@pytest.mark.parametrize("name,kw,precision", [("ch11a", {}, 0), ("ch11a", {}, 1), ("teapot", {"lite": True}, 0), ("dice", {"nx": 2, "ny": 1}, 0)],
                         ids=["ch11a-f32", "ch11a-f64", "teapot-lite-f32", "dice-2x1-f32"])
def test_passes_cost_only_the_samples_they_add(name, kw, precision):
    """rtmi_progressive_* through the C ABI: the per-pixel sums stay on the device between passes.  The frame after the pass
    with cap c is rtmi_render's frame with max_samples = c, bit for bit, and all passes together take exactly the samples -
    and trace exactly the rays - of one render with the last cap (here: none, camera.rs:290-299 runs to its end)."""
    from raytracing_amd import abi
    world, cam = scenes.BASELINE_SCENES[name](200, 104, **kw)
    sc = abi.Scene(world)
    caps = [1, 3, 16, 17, 64, 0]
    prog = abi.Progressive(sc, cam, abi.make_opts(seed=9, precision=precision))
    total = {k: 0 for k in ("primary", "reflection", "refraction", "shadow", "samples")}
    for c in caps:
        img, st = prog.render_pass(c)
        ref, rst = abi.render_host(sc, cam, abi.make_opts(seed=9, precision=precision, max_samples=c))
        assert np.array_equal(img, ref), c
        for k in total:
            total[k] += st[k]
        assert st["overflow"] == 0
    # a pass with nothing left to do takes nothing
    img2, st2 = prog.render_pass(0)
    assert np.array_equal(img2, img) and st2["samples"] == 0
    prog.close()
    for k in total:
        assert total[k] == rst[k], k
    sc.close()
