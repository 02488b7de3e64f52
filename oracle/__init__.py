"""CPU oracle for the FlowNet2 hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of what fperezgamonal/flownet2-tf computes on the path
image pair -> FlowNetS/C/SD/CS/CSS/2 forward -> .flo.  Every
function cites the reference file:line it follows.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline
only.  The product (``flownet2-tf_amd/``) never imports it and has no CPU
fallback: it fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``flowio`` (.flo read/write, EPE, Middlebury colour coding) is PINNED against
    the reference's own ``src/flowlib.py`` run in the build container
    (tests/golden/make_golden_flowlib.py -> tests/golden/flowlib_*.npz) and against
    the reference's sample ``.flo`` files.
  * ``ops`` (correlation / flow_warp / downsample and their gradients): the
    reference kernels are CUDA + TensorFlow-plugin code registered for
    DEVICE_GPU only; they can be neither compiled nor run here and the
    reference holds no known-answer vectors for them => PARITY UNPINNED.
    Mitigation: two independent restatements (a literal per-thread loop
    transliteration of the kernel indexing and a vectorised one) are checked
    against each other.
  * ``nn`` / ``models`` (conv, conv-transpose, resize, the model graphs): the
    arithmetic lives in TensorFlow 1.x + cuDNN (not vendored, not installed)
    => PARITY UNPINNED; the restatement of the TF op definitions is
    cross-checked against torch's CPU kernels as an independent implementation.
  * ``train`` (FlowNetS loss gradients by torch float64 autograd over the restated graph, TF-form Adam): same
    status as ``models`` => PARITY UNPINNED; its torch forward is tested against the NumPy forward to 1e-9.
  * ``augment`` (DataAugmentation / FlowAugmentation of the preprocessing plugin): the plugin needs TensorFlow
    headers to build and has no known-answer vectors => PARITY UNPINNED; two forms (vectorised, literal loops)
    checked against each other plus identity / inverse properties.
  * The MPI-Sintel metric code has no oracle copy: the product's ``src/flowlib.py`` is tested directly against
    outputs of the reference's own functions (tests/golden/make_golden_metrics.py -> metrics_golden.npz): PINNED.
"""
