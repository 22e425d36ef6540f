"""CPU oracle for the depth-soft captioning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / timed CPU baseline.
The product path (``depth_image_captioning_pub_amd``) never imports this
package and fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * decoder (soft + hard), Soft/Hard attention, depth CNN encoder, loss,
    greedy decode: PINNED against golden vectors captured from the imported
    reference classes (tests/golden/make_golden.py ran in the build container
    against /root/reference; the vectors are committed under tests/golden/).
  * ResNet-152 RGB encoder (torchvision, un-vendored, version unpinned in the
    reference) : PARITY UNPINNED - restated from the public torchvision
    Bottleneck-v1.5 definition, cross-checked only against torch's own CPU
    conv/batch_norm ops.
"""
