"""Trainers with the reference's class names and train_step() surface: SimCLR, BYOL, BarlowTwins, DINO, SimSiam, ReLIC, MoCo."""
