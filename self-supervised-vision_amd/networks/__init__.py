"""Encoders of the accelerated path: ResNets (resnet.py) and the reference's Transformer encoder (vit.py), composed from the HIP ops."""
