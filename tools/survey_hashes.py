"""SURVEY 8c quotes four hashes "from the reference source compiled here" that round 1 could not reproduce.  This script
evaluates the definition the survey states — byte-wise FNV-1a-64 (offset 0xcbf29ce484222325, prime 0x100000001b3) over the raw
little-endian fp32 arrays; for the hit map `h = (h ^ uint32(idx)) * prime` per pixel, rows bottom-up, miss = -1 — on the oracle's
arrays (which the product's host layer equals bit for bit) and prints what it gives, plus the obvious alternative readings.
None of them matches, although every count / known answer the survey lists beside the hashes does (4970 / 1741, 19070 / 6680,
45 494 hits, triangle 2054 at 0.76238, cache[23] = 5.94006e-07).  The HDR float array is the telling case: RGBE decoding
(hdrloader.cpp:99-104, val/256 * 2^(e-128)) has no freedom at all — max texel 21.0, mean 0.30 as the survey says — so a
different hash of the same floats means the recorded value or its definition is off, not the data.  Nothing is pinned by this.
usage: python tools/survey_hashes.py > profiles/r02_survey_hashes.log"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_lib as O
from ray_tracing_amd import scenes, ASSETS

SURVEY = {"chap5 triangle positions (p1,p2,p3; 9 floats/tri; final order)": 2007996619683796983,
          "HDR float data": 9689102154804598586, "HDR cache": 55182048846039717, "1080p hit map, eye (0,0,4)": 6796969391178241497}
M = (1 << 64) - 1
OFF, PRIME = 0xcbf29ce484222325, 0x100000001b3


def fnv1a_bytes(b):
    import ray_tracing_amd as rt
    return rt.hash_fnv1a64(np.frombuffer(b, np.uint8))          # byte-wise FNV-1a-64 (KATs in tests/test_host_oracle.py)


def fnv1_bytes(b):
    h = OFF
    for chunk in np.array_split(np.frombuffer(b, np.uint8), max(1, len(b) // (1 << 16))):
        for v in chunk.tolist():
            h = ((h * PRIME) & M) ^ v
    return h


def fnv1a_words(a):
    h = OFF
    for v in np.ascontiguousarray(a).view(np.uint32).reshape(-1).tolist():
        h = ((h ^ v) * PRIME) & M
    return h


def report(name, arr, small=False):
    raw = np.ascontiguousarray(arr, np.float32).tobytes()
    print(f"{name}: {len(raw)} bytes; survey recorded {SURVEY[name]}")
    print(f"    byte-wise FNV-1a-64 (the stated definition) = {fnv1a_bytes(raw)}")
    print(f"    word-wise FNV-1a-64 over uint32            = {fnv1a_words(np.frombuffer(raw, np.uint32))}")
    if small:
        print(f"    byte-wise FNV-1 (multiply first)           = {fnv1_bytes(raw)}")
    print(f"    match: {SURVEY[name] in (fnv1a_bytes(raw), fnv1a_words(np.frombuffer(raw, np.uint32)))}")


m = O.Mesh()
mat = np.zeros(18, np.float32); mat[3:6] = (1.0, 0.84, 0.0); mat[6:18] = [0, 1.0, 0.8, 0, 0.6, 0, 0, 0, 0.2, 0.2, 1, 0]
m.read_obj(os.path.join(ASSETS, "models", "nanosuit.obj"), mat, O.transform_matrix((0, 0, 0), (-0.8, 0.2, 0), (3, 3, 3)), True)
mat2 = mat.copy(); mat2[3:6] = 1.0; mat2[10] = 0.05; mat2[14] = 0.2; mat2[7] = 0.0; mat2[15] = 1.0
m.read_obj(os.path.join(ASSETS, "models", "quad.obj"), mat2, O.transform_matrix((0, 0, 0), (0, 0.2, 0), (18.83, 0.01, 18.83)), False)
nodes = m.build(8, True)
tris = m.triangles()
pt, pn = scenes.chap5_scene()
print(f"chap5 scene: {tris.shape[0]} triangles, {nodes.shape[0]} nodes (survey: 19070 / 6680); oracle == product host layer: "
      f"{np.array_equal(tris.view(np.uint32), pt.view(np.uint32)) and np.array_equal(nodes[1:].view(np.uint32), pn[1:].view(np.uint32))}")
report("chap5 triangle positions (p1,p2,p3; 9 floats/tri; final order)", tris[:, :9])
hdr = O.hdr_load(scenes.HDR_PATH)
print(f"HDR: {hdr.shape[1]}x{hdr.shape[0]}, max texel {hdr.max():g}, mean {hdr.mean():.4f} (survey: 1024x512, 21.0, 0.30)")
report("HDR float data", hdr)
cache = O.hdr_cache(hdr)
print(f"cache[23] = {cache.reshape(-1)[23]:.6g} (survey: 5.94006e-07)")
report("HDR cache", cache)
t2, n2 = scenes.chap2_scene()
tri, dist = O.primary_hits(t2, n2, 1920, 1080, np.array([0, 0, 4], np.float32), np.eye(4, dtype=np.float32).reshape(16))
print(f"hit map: {(tri >= 0).sum()} hits (survey: 45494)")
for label, a in (("rows bottom-up (as stored)", tri), ("rows top-down", tri[::-1])):
    print(f"    per-pixel FNV-1a over uint32(idx), {label} = {O.hitmap_hash(a)}   (survey {SURVEY['1080p hit map, eye (0,0,4)']})")
