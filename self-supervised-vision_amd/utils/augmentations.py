"""GPU augmentation front-end (placeholder until the fused kernels land in this round)."""


def get_transform(config):
    raise NotImplementedError("GPU augmentation kernels are not built yet")
