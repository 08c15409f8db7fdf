"""Hand-built scenes shared by the GPU tests."""
import numpy as np


def deep_chain(depth, base_color=1.0, roughness=0.0):
    """A left-deep chain of `depth` internal nodes (left = the rest of the chain, right = a one-triangle leaf) whose deep
    subtree is always the nearer child for an eye at z = +2: the traversal stack grows to `depth` entries, past the 18 kept
    in LDS, so the global spill part of the stack is exercised.  Returns (Triangle_encoded[depth+1], BVHNode_encoded[2*depth+2])."""
    tris = np.zeros((depth + 1, 36), np.float32)
    for k in range(depth + 1):
        z = -1.0 + 0.01 * k  # higher index = nearer to the eye
        tris[k, :9] = np.array([[-0.5, -0.5, z], [0.5, -0.5, z], [0.0, 0.5, z]], np.float32).reshape(-1)
        tris[k, 9:18] = np.tile([0, 0, 1], 3)
        tris[k, 21:24] = base_color
        tris[k, 28] = roughness
        tris[k, 34] = 1.0

    def box(lo, hi):
        pts = tris[lo:hi, :9].reshape(-1, 3)
        return pts.min(0), pts.max(0)

    nodes = np.zeros((2 * depth + 2, 12), np.float32)
    ids = [1 + 2 * i for i in range(depth)]
    last_leaf = 1 + 2 * depth
    for i in range(depth):
        me, leaf = ids[i], ids[i] + 1
        nxt = ids[i + 1] if i + 1 < depth else last_leaf
        nodes[me, 0], nodes[me, 1] = nxt, leaf
        nodes[me, 6:9], nodes[me, 9:12] = box(i, depth + 1)
        nodes[leaf, 3], nodes[leaf, 4] = 1, i
        nodes[leaf, 6:9], nodes[leaf, 9:12] = box(i, i + 1)
    nodes[last_leaf, 3], nodes[last_leaf, 4] = 1, depth
    nodes[last_leaf, 6:9], nodes[last_leaf, 9:12] = box(depth, depth + 1)
    return tris, nodes
