"""MI355X-native FlowNet2 engine behind the op / entry-point surface of
fperezgamonal/flownet2-tf's ``src`` package.

Run from ``flownet2-tf_amd/`` exactly like the reference:
``python -m src.flownet_s.test --input_a A --input_b B --out DIR``.
"""
