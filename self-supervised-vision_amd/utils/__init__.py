"""Host-side helpers mirroring the reference's utils package: losses, optimiser / scheduler factories, GPU data path, evaluation."""
