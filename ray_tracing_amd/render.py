"""Command-line driver: what the reference's main() + display() loop do for one still image (c5:795-933, 706-760).

    python -m ray_tracing_amd.render --config 5 --frames 256 --out chap5.png

config 5 = path_tracing_5chap (nanosuit + floor, Disney BRDF + HDR importance sampling + MIS), config 4 = path_tracing_4chap
(sphere2, anisotropic Disney BRDF, uniform hemisphere sampling), config 3 = the chapter-3 diffuse loop on the chapter-5
scene, config 2 = chapter 2's closest-hit query shown as a hit/depth image, config 1 = chapter 1's CPU Cornell box.
`--dump-every N` writes a PNG every N frames, like the per-spp dumps of c1:478-482.
"""
from __future__ import annotations

import argparse
import time

import numpy as np

import ray_tracing_amd as rt
from ray_tracing_amd import scenes


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", type=int, default=5, choices=[1, 2, 3, 4, 5])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--frames", type=int, default=64, help="frameCounter range 0..frames-1 (1 spp each); chap1: SAMPLE")
    ap.add_argument("--rotate", type=float, default=0.0, help="rotatAngle of c5:716 (degrees)")
    ap.add_argument("--up", type=float, default=20.0, help="upAngle of c5:716 (degrees); the reference starts at 0")
    ap.add_argument("--r", type=float, default=4.0, help="orbit radius, c5:149")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--dump-every", type=int, default=0)
    ap.add_argument("--out", default="out.png")
    a = ap.parse_args(argv)

    if a.config == 1:
        W, H = a.width or 256, a.height or 256
        t0 = time.perf_counter()
        img = rt.chap1_render(W, H, a.frames)
        rt.write_png(a.out, rt.chap1_to_rgb8(img))
        print(f"chap1 {W}x{H} x {a.frames} samples: {time.perf_counter() - t0:.2f} s -> {a.out}")
        return 0

    W, H = a.width or 1920, a.height or 1080
    eye, cam = rt.camera(a.rotate, a.up, a.r)
    if a.config == 2:
        tris, nodes = scenes.chap2_scene()
        sc = rt.Scene(tris, nodes, device=a.device)
        tri, dist = sc.primary_hits(rt.make_desc(W, H, eye=eye, camera_rotate=cam))
        hit = tri >= 0
        shade = np.zeros((H, W, 3), np.float32)
        if hit.any():
            dn, df = dist[hit].min(), dist[hit].max()
            shade[hit] = (1.0 - 0.8 * (dist[hit] - dn) / max(df - dn, 1e-6))[:, None]
        rt.write_png(a.out, (np.clip(shade[::-1], 0, 1) * 255 + 0.5).astype(np.uint8))
        print(f"chap2 closest hits {W}x{H}: {int(hit.sum())} of {hit.size} rays hit -> {a.out}")
        return 0

    tris, nodes = scenes.chap4_scene() if a.config == 4 else scenes.chap5_scene()
    hdr, cache = scenes.hdr_and_cache()
    sc = rt.Scene(tris, nodes, hdr, cache, device=a.device)
    desc = rt.make_desc(W, H, mode=a.config, eye=eye, camera_rotate=cam)
    img = np.zeros((H, W, 3), np.float32)  # lastFrame
    step = a.dump_every if a.dump_every > 0 else a.frames
    t0 = time.perf_counter()
    done = 0
    while done < a.frames:
        n = min(step, a.frames - done)
        img = sc.render(desc, done, n, img)  # frames done..done+n-1 folded into the running mean (f5:954-955)
        done += n
        if a.dump_every > 0 and done < a.frames:
            rt.write_png(f"{a.out}.{done:05d}.png", rt.tonemap_rgb8(img))
    dt = time.perf_counter() - t0
    rt.write_png(a.out, rt.tonemap_rgb8(img))  # pass3: tone map + gamma (pass3.fsh:14-25)
    print(f"chap{a.config} {W}x{H} x {a.frames} spp: {dt:.2f} s incl. host copies ({W * H * a.frames / dt / 1e6:.0f} Mpaths/s) -> {a.out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
