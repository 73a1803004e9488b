"""Host logic of the set-abstraction sampling chain (no GPU): the record that tells a module its
cloud is the previous module's centroids in pick order (dropin/pointnet2/pointnet2_utils.py:
remember_head / head_record) is keyed on the tensor's identity and in-place version."""
import importlib

import torch

importlib.import_module("3dioumatch_amd")
utils = importlib.import_module("pointnet2.pointnet2_utils")


def test_head_record_follows_identity_and_version():
    new_xyz = torch.rand(2, 16, 3)
    ties = torch.tensor([16, 16], dtype=torch.int32)
    assert utils.head_record(new_xyz) is None
    utils.remember_head(new_xyz, ties)
    assert utils.head_record(new_xyz) is ties
    assert utils.head_record(new_xyz.clone()) is None          # a copy is another cloud
    assert utils.head_record(new_xyz[:, :8]) is None           # so is a slice
    assert utils.head_record(new_xyz.transpose(1, 2)) is None
    new_xyz.add_(1.0)                                           # modified in place: stale
    assert utils.head_record(new_xyz) is None
    utils.remember_head(new_xyz, None)                          # nothing to remember
    assert utils.head_record(new_xyz) is None


def test_head_record_rejects_a_mismatched_record():
    new_xyz = torch.rand(3, 8, 3)
    utils.remember_head(new_xyz, torch.tensor([8, 8], dtype=torch.int32))   # wrong batch size
    assert utils.head_record(new_xyz) is None


def test_head_records_do_not_outlive_their_tensors():
    before = len(utils._HEADS)
    for _ in range(200):
        t = torch.rand(1, 4, 3)
        utils.remember_head(t, torch.tensor([4], dtype=torch.int32))
    assert len(utils._HEADS) <= before + 66    # dead entries are swept once the table passes 64
