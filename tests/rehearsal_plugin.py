"""Stand-in renderer for bench.py's CPU rehearsal (`--backend gloo --render-plugin rehearsal_plugin:make`): the oracle in SUM
mode on the small scene, so the launcher / partition / reduce / total / verify path of bench.py can run with 2 gloo ranks on
a machine without GPUs.  Test infrastructure: bench.py never loads it on the GPU path."""
import numpy as np

import oracle_lib as O


class _OracleRenderer:
    def __init__(self, args, rank, world):
        import ray_tracing_amd as rt
        from ray_tracing_amd import scenes
        self.tris, self.nodes = scenes.small_scene()
        self.hdr, self.cache = scenes.hdr_and_cache()
        self.eye, self.cam = rt.camera(10.0, 12.0, 3.2)
        self.W, self.H, self.mode = args.width, args.height, args.mode

    def render_sum(self, frame_begin, frame_count):
        acc, _ = O.render(self.tris, self.nodes, self.hdr, self.cache, self.W, self.H, self.mode, self.eye, self.cam,
                          frame_begin, frame_count, accumulate=1)
        return np.asarray(acc, np.float32)


def make(args, rank, world):
    return _OracleRenderer(args, rank, world)
