"""Two host threads in the rasterizer forward at once.  The per-call host state of the C ABI is thread-local: the instance
count's mail word and counter (csrc/api.hip, CountMail: K1's last block stores the count into host-coherent memory the calling
thread polls), the profiler aside.  Each thread must get ITS frame's count and images, whatever the interleaving of the launches
on the (shared) stream."""
import threading

import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def test_two_threads_get_their_own_counts():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    scenes = [util.scene_inputs(4000 + 700 * k, 640, 480, scene_seed=61 + k, cam_seed=62 + k) for k in range(4)]
    ref = []
    for inp in scenes:
        out, _ = util.hip_forward_raw(inp, "FTT")
        ref.append((out[0], out[1].clone()))
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            for it in range(12):
                k = (tid + 2 * it) % 4 if it % 3 else (3 * tid + it) % 4
                out, _ = util.hip_forward_raw(scenes[k], "FTT")
                torch.cuda.synchronize()
                if out[0] != ref[k][0]:
                    errors.append("thread %d iteration %d scene %d: num_rendered %d, expected %d" % (tid, it, k, out[0], ref[k][0]))
                elif not torch.equal(out[1], ref[k][1]):
                    errors.append("thread %d iteration %d scene %d: image differs" % (tid, it, k))
        except Exception as e:  # noqa: BLE001  (reported below, in the main thread)
            errors.append("thread %d: %r" % (tid, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"
    assert not errors, errors[:5]
