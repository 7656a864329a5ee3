"""GPU: kNN evaluation kernel (ssv_knn_label_agreement) against the reference's compute_neighbor_accuracy goldens and the oracle."""
import numpy as np
import pytest
import torch

from oracle import evalknn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_knn_accuracy_matches_reference_goldens(dev, golden):
    from ssv_amd.utils import eval_utils
    g = golden["eval_level"]
    for (seed, n, d, classes, spread, k), want in zip(g["knn_cases"], g["knn_accuracy"]):
        fvecs, labels = evalknn.clustered_features(int(seed), int(n), int(d), int(classes), float(spread))
        got = eval_utils.compute_neighbor_accuracy(fvecs, labels, k=int(k))
        # the n x k table is integer; a rounding-level near-tie may swap one boundary neighbour
        assert abs(got - want) <= 2.0 / (n * k), (seed, got, want)


def test_knn_counts_bit_exact_on_exactly_representable_scores(dev):
    """Small-integer features: every inner product is exact in fp32, so ordering (with ties by index) is fully determined."""
    from ssv_amd import ops
    g = torch.Generator().manual_seed(5)
    for n, d, k in ((300, 16, 20), (1000, 8, 7), (65, 4, 40), (4100, 12, 20), (21, 4, 20)):
        z = torch.randint(-3, 4, (n, d), generator=g).float()
        labels = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32)
        want = evalknn.neighbor_agreement_count(z.numpy(), labels.numpy(), k)
        assert ops.knn_label_agreement(z.to(dev), labels.to(dev), k) == want, (n, d, k)


@pytest.mark.parametrize("n,d,k", [(21, 32, 20), (65, 64, 20), (300, 128, 20), (1000, 32, 7), (4100, 64, 20), (4100, 128, 1), (12000, 32, 20), (9000, 128, 20)])
def test_fused_knn_counts_bit_exact_on_exactly_representable_scores(dev, n, d, k):
    """The search fused into the bf16x3 Gram product (d in {32, 64, 128}, k <= 20; csrc/evalknn.hip knn_fused_k, one part up to the merge of several column parts at the
    larger n): small-integer features make every score exact and full of ties, so the count is fully determined by value-descending / index-ascending order."""
    from ssv_amd import ops
    g = torch.Generator().manual_seed(n + d + k)
    z = torch.randint(-2, 3, (n, d), generator=g).float()
    labels = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32)
    want = evalknn.neighbor_agreement_count(z.numpy(), labels.numpy(), k)
    with ops.arithmetic("bf16x3"):
        assert ops.knn_label_agreement(z.to(dev), labels.to(dev), k) == want
    with ops.arithmetic("f32"):
        assert ops.knn_label_agreement(z.to(dev), labels.to(dev), k) == want


def test_fused_knn_all_scores_equal_and_duplicates(dev):
    """Every score the same (identical features): the neighbours of every query are the lowest indices, and the best hit dropped is index 0 - not the query."""
    from ssv_amd import ops
    n, d, k = 700, 64, 20
    z = torch.ones(n, d)
    labels = (torch.arange(n) % 3).to(torch.int32)
    want = evalknn.neighbor_agreement_count(z.numpy(), labels.numpy(), k)
    assert ops.knn_label_agreement(z.to(dev), labels.to(dev), k) == want


def test_fused_knn_agrees_with_the_unfused_search_on_unit_features(dev):
    """Random unit features (the reference's input, proj_dim 128): the fused bf16x3 search, the fp32-MFMA Gram + selection and the fp64 oracle give the same count up to
    rounding-level near-ties at the 21st place (a handful in n k = 10^5 pairs)."""
    from ssv_amd import ops
    n, d, k = 5000, 128, 20
    g = torch.Generator().manual_seed(77)
    z = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)
    labels = torch.randint(0, 10, (n,), generator=g, dtype=torch.int32)
    want = evalknn.neighbor_agreement_count(z.numpy(), labels.numpy(), k)
    with ops.arithmetic("bf16x3"):
        fused = ops.knn_label_agreement(z.to(dev), labels.to(dev), k)
        again = ops.knn_label_agreement(z.to(dev), labels.to(dev), k)
    with ops.arithmetic("f32"):
        unfused = ops.knn_label_agreement(z.to(dev), labels.to(dev), k)
    assert fused == again                               # appends in a fixed order: reproducible
    assert abs(fused - want) <= 4 and abs(unfused - want) <= 4, (fused, unfused, want)


def test_knn_large_set_multi_chunk_property(dev):
    """n above one S chunk (4096 rows): planted duplicates - every vector appears exactly twice with the same label and is far
    from everything else, so with k=1 (after dropping the best hit) the agreement count is exactly n."""
    from ssv_amd import ops
    n_half, d = 6000, 64
    g = torch.Generator().manual_seed(9)
    base = torch.nn.functional.normalize(torch.randn(n_half, d, generator=g), dim=1)
    z = torch.cat([base, base])
    labels = torch.arange(n_half, dtype=torch.int32).repeat(2)
    assert ops.knn_label_agreement(z.to(dev), labels.to(dev), 1) == 2 * n_half


def test_knn_rejects_bad_arguments(dev):
    from ssv_amd import _lib, ops
    z = torch.randn(10, 8, device=dev)
    labels = torch.zeros(10, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.SsvError):
        ops.knn_label_agreement(z, labels, 10)          # k must be < n
    with pytest.raises(_lib.SsvError):
        ops.knn_label_agreement(z.cpu(), labels.cpu(), 3)


def test_linear_probe_matches_oracle(dev):
    """The GPU probe (MFMA logits / weight gradient, softmax-CE kernel, plain-momentum SGD kernel) against the torch CPU restatement
    with the same init draw and the same batch order."""
    from ssv_amd.utils import eval_utils
    xtr, ytr = evalknn.clustered_features(81, 700, 126, 10, 2.0)             # 126: not a multiple of 4 -> padded; 700: ragged last batch
    xte, yte = evalknn.clustered_features(81, 300, 126, 10, 2.0)
    cfg = {"epochs": 6, "batch_size": 128, "lr": 0.1, "input_dim": 128}
    torch.manual_seed(5)
    want, head = evalknn.linear_evaluation(cfg, {"fvecs": xtr, "labels": ytr}, {"fvecs": xte, "labels": yte}, 10, eval_utils.probe_batches)
    torch.manual_seed(5)
    got = eval_utils.linear_evaluation(cfg, {"fvecs": xtr, "labels": ytr}, {"fvecs": xte, "labels": yte}, 10, dev)
    assert abs(got - want) <= 1.0 / 300 + 1e-9, (got, want)
    assert want > 0.5                                                         # the probe actually learns the planted classes


def test_softmax_ce_kernel(dev):
    from ssv_amd import _lib
    n, c, ld = 37, 10, 12
    logits = torch.randn(n, ld, generator=torch.Generator().manual_seed(3)) * 3
    labels = torch.randint(0, c, (n,), generator=torch.Generator().manual_seed(4), dtype=torch.int32)
    ref_in = logits[:, :c].double().requires_grad_()
    ref = torch.nn.functional.nll_loss(torch.log_softmax(ref_in, -1), labels.long())
    ref.backward()
    lg, lb = logits.to(dev), labels.to(dev)
    stats, dl = torch.zeros(2, device=dev), torch.full((n, ld), 7.0, device=dev)
    ws = torch.empty(_lib.load().ssv_softmax_ce_workspace_bytes(n), dtype=torch.uint8, device=dev)
    _lib.call("ssv_softmax_ce_fwd_bwd", n, c, ld, _lib.ptr(lg), _lib.ptr(lb), _lib.ptr(stats), _lib.ptr(dl), _lib.ptr(ws), ws.numel(), _lib.stream())
    np.testing.assert_allclose(stats[0].item(), ref.item(), rtol=1e-6)
    np.testing.assert_allclose(stats[1].item(), float((ref_in.argmax(-1) == labels.long()).float().mean()), rtol=1e-6)
    np.testing.assert_allclose(dl[:, :c].cpu().numpy(), ref_in.grad.numpy(), rtol=1e-5, atol=1e-8)
    assert float(dl[:, c:].abs().max()) == 0.0
