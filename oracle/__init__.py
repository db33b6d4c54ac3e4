"""CPU oracle for the tf_face_toolbox_amd hot path.  TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the arithmetic that the reference
(medivhna/TF_Face_Toolbox) instantiates for its data-parallel training step
(`data_parallel.py`, `nets/sphere.py`, `nets/net_base.py`, `loss.py`,
`train.py:122-144`).  It exists to CHECK the HIP path; it is never the thing
shipped or measured.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  Nothing under
`tf_face_toolbox_amd/` imports it, and the product path raises when the HIP
extension is missing instead of falling back to this code.

PARITY UNPINNED.  The reference is Python 2 + TensorFlow 1.x (`tf.contrib`),
cannot be imported or executed in this image (no python2, no tensorflow, and
`data_parallel.py:19` imports a module that is not in the tree), and ships no
tests, fixtures or golden vectors.  The
arithmetic itself lives in un-vendored, un-pinned TensorFlow ("r1.8 or above",
README.md:4-6).  The oracle therefore restates TF-1.x's documented op
semantics at the reference's call sites, and is
cross-validated in `tests/test_oracle.py` against an independent second
implementation (torch-CPU float64 `conv2d` with explicit TF-SAME padding and
autograd) and against central finite differences.  The A-softmax head is not
in the reference snapshot at all (README.md:14,19 only claims it); it follows
the SphereFace paper.
"""
