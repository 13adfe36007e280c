"""The host-side tree builder (binned SAH, folded to four children per node, renumbered) checked on the CPU
through rt_bvh_build_check: every leaf referenced exactly once with its own box, inner boxes cover what lies
beneath them, parents are numbered before their children.  Replaces the pointer tree of Hittable::bvh
(src/hittable.rs:103-155), which the reference only ever checks by rendering."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import _abi


def check(boxes):
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    rep = _abi.RtBvhReport()
    rc = _abi.lib_amd().rt_bvh_build_check(boxes.ctypes.data_as(C.POINTER(C.c_float)), len(boxes), C.byref(rep))
    assert rc == 0, _abi.lib_amd().rt_last_error().decode()
    return rep


def random_boxes(rng, n, extent=100.0, size=1.0):
    c = rng.uniform(-extent, extent, (n, 3))
    h = rng.uniform(0.01 * size, size, (n, 3))
    return np.concatenate([c - h, c + h], axis=1)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 7, 64, 1000, 30000])
def test_every_leaf_once_and_boxes_nested(n):
    rep = check(random_boxes(np.random.default_rng(n), n))
    assert rep.violations == 0 and rep.leaves_found == n
    assert rep.width in (2, 4)
    if n >= 2:
        assert 1 <= rep.nodes <= n - 1
        assert rep.depth <= 40 + 24   # SAH splits, median splits below depth 40
        assert 1 <= rep.stack_need <= (rep.width - 1) * rep.depth
        assert rep.stack_need <= 64, "random boxes must fit the traversal stack"


def test_degenerate_inputs_still_give_a_valid_tree():
    rng = np.random.default_rng(1)
    same = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (500, 1))        # all boxes identical: SAH cannot split
    assert check(same).violations == 0
    line = random_boxes(rng, 2000)
    line[:, [1, 2, 4, 5]] = 0.0                                                  # flat boxes on a line
    assert check(line).violations == 0
    huge = random_boxes(rng, 300, extent=1e6, size=5e5)
    huge[::7] = random_boxes(rng, len(huge[::7]), extent=1.0, size=1e-3)         # tiny boxes among huge ones
    assert check(huge).violations == 0


def test_chain_shaped_tree_stays_within_the_traversal_stack():
    """Boxes that double in size and distance make SAH peel one leaf off per level: the deepest tree f32 coordinates
    allow.  Its worst-case stack need stays within the 64 entries the kernels have (rt_scene_create checks this number)."""
    x = np.array([2.0 ** i for i in range(120)])
    h = 0.25 * x
    rep = check(np.stack([x - h, -h, -h, x + h, h, h], axis=1))
    assert rep.violations == 0 and rep.leaves_found == 120
    assert rep.depth >= 15 and rep.stack_need <= 64


def test_outliers_on_top_of_a_cluster_are_rebuilt_with_a_shallower_sah_part():
    """Sixty outliers at tripling distances over a cluster of a thousand small boxes: with SAH down to depth 40 the
    one-versus-rest chain plus the cluster's own levels would need 78 stack entries; the builder then leaves less of
    the tree to SAH until the worst case fits the 64 entries of the traversal stack."""
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 1, (1000, 3))
    cluster = np.concatenate([c - 0.001, c + 0.001], axis=1)
    k = np.arange(1, 61, dtype=np.float64)
    far = np.zeros((60, 6))
    far[:, 0], far[:, 3], far[:, 4], far[:, 5] = 3.0 ** k, 3.0 ** k + 1, 1, 1
    rep = check(np.concatenate([cluster, far]))
    assert rep.violations == 0 and rep.leaves_found == 1060
    assert rep.stack_need <= 64


def test_sah_tree_is_cheaper_than_a_shuffled_insertion_order_would_suggest():
    """The surface-area cost of the tree does not depend on the order the leaves are handed in."""
    rng = np.random.default_rng(3)
    boxes = random_boxes(rng, 5000)
    a = check(boxes)
    b = check(boxes[rng.permutation(len(boxes))])
    assert a.violations == 0 and b.violations == 0
    assert abs(a.sah_cost - b.sah_cost) <= 0.02 * a.sah_cost


def test_parallel_build_of_a_large_tree():
    rep = check(random_boxes(np.random.default_rng(9), 300000, extent=500.0, size=0.5))
    assert rep.violations == 0 and rep.leaves_found == 300000
