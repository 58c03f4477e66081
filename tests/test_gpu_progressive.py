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
