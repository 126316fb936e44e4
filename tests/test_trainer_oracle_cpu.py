"""CPU: the trainer's update rule pinned to the REFERENCE: tests/golden/trainer.npz holds what the
reference's own build_optimizer / optimizer_step / ModelEMA produce; checked here are (1) the oracle's written-out rule
(oracle/trainer_ref.py) and (2) the PRODUCT's trainer host logic on CPU tensors (flat buffers and per-tensor layout).  The same
comparison runs on the MI355X with the fused optimizer kernels in tests/test_trainer_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import trainer_ref as T, yolo11_ref as R
from tests._golden import check, load
from tests._trainer_parity import compare_with_reference, product_trainer, run_three_steps, tiny_sd


@pytest.mark.parametrize("tag", ["sgd", "auto"])
def test_oracle_update_rule_matches_reference(tag):
    gold = load("trainer.npz")
    sd = tiny_sd()
    trainable = [str(n) for gi in (0, 1, 2) for n in gold[f"{tag}.group{gi}.names"] if ".dfl" not in str(n)]
    order = [k for k in sd if k in set(trainable)]                          # named_parameters order == state_dict order
    norm = {str(n) for n in gold[f"{tag}.group2.names"]}                    # reference groups: [bias, decay, norm]
    assert all(k.endswith("bn.weight") for k in norm)
    assert all("bias" in str(n) for n in gold[f"{tag}.group0.names"])
    name, lr, mom = ("SGD", 0.01, 0.937) if tag == "sgd" else T.auto_optimizer(4, 300)
    assert name == str(gold[f"{tag}.optimizer"])
    st = T.RefTrainerState(sd, order, norm, name=name, lr=lr, momentum=mom, decay=5e-4, ema_updates=T.EMA_START_UPDATES)
    # the restated grouping equals the reference's
    dg, ng, bg = T.param_groups([(k, None) for k in order], norm)
    # (membership: the order inside a group is the module registration order, which the flat state_dict does not carry)
    assert set(bg) == {str(n) for n in gold[f"{tag}.group0.names"]}
    assert set(dg) == {str(n) for n in gold[f"{tag}.group1.names"] if ".dfl" not in str(n)} and set(ng) == {str(n) for n in gold[f"{tag}.group2.names"]}
    for step in range(3):
        grads = {k: T.synthetic_grad(k, sd[k].shape, step) for k in order}
        T.perturb_buffers(st.sd, step)
        st.optimizer_step(grads)
    for which, got in (("model", st.sd), ("ema", st.ema)):
        for k, (nrm, sm) in zip([str(n) for n in gold[f"{tag}.{which}.names"]], gold[f"{tag}.{which}.norm_sum"]):
            v = got[k].double()
            assert abs(v.norm().item() - nrm) <= 2e-5 * max(nrm, 1e-6), (which, k, v.norm().item(), nrm)
        for k in ("model.0.conv.weight", "model.0.bn.running_var", "model.23.cv3.2.2.weight", "model.23.cv2.0.2.bias"):
            check(gold, f"{tag}.{which}.{k}", got[k], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("flat", [True, False])
@pytest.mark.parametrize("tag", ["sgd", "auto"])
def test_product_trainer_update_rule_matches_reference_on_cpu(tag, flat):
    tr = run_three_steps(product_trainer(tag, "cpu", flat=flat))
    compare_with_reference(tr, tag, rtol=2e-5)
