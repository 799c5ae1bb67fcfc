"""Joint-error read-out of the estimate modes (reference: src/depth_train.py:200-253 and
src/utils/handpose_evaluation.py:92-97,130-136,197-203)."""
import numpy as np

NYU_EVAL_JOINTS = np.array([0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32])    # depth_train.py:232


class HandposeEvaluation(object):
    """gt / joints: [frames, joints, 3] in mm."""

    def __init__(self, gt, joints):
        self.gt, self.joints = np.asarray(gt, np.float64), np.asarray(joints, np.float64)
        self.err = np.sqrt(np.square(self.gt - self.joints).sum(axis=2))          # [frames, joints]

    def getMeanError(self):                          # handpose_evaluation.py:92-97
        return float(np.nanmean(np.nanmean(self.err, axis=1)))

    def getMaxErrorOverSeq(self):                    # :130-136
        return np.nanmax(self.err, axis=1)

    def getWorstJoint(self):
        return np.argmax(self.err, axis=1)

    def getNumFramesWithinMaxDist(self, dist):       # :197-203
        return int((np.nanmax(self.err, axis=1) <= dist).sum())


def to_mm(pose, com, cube, nyu=True):
    """[n, J*3] normalised pose -> [n, J', 3] mm (depth_train.py:231-239)."""
    pose = np.asarray(pose, np.float32)
    n = pose.shape[0]
    p = pose.reshape(n, -1, 3)
    if nyu:
        p = p[:, NYU_EVAL_JOINTS]
    return p * (np.asarray(cube, np.float32)[0] / 2.0) + np.asarray(com, np.float32).reshape(n, 1, 3)


def evaluate(trainer, batches, mode_idx, nyu=True):
    """batches: iterable of (images_b, labels_b, com_b, cube) device tensors / arrays.
    Returns (mean_err_mm, pct_frames_within_40mm) exactly as the driver prints them (depth_train.py:248-253)."""
    import torch
    trainer.dis.eval()
    gt3d, pr3d = [], []
    with torch.no_grad():
        for images, labels, com, cube in batches:
            regress = trainer.dis.regress_a if mode_idx == 0 else trainer.dis.regress_b      # :200-203
            _, post, _ = regress(images)
            pose = trainer.vae.decode(post)
            com_np = com.detach().cpu().numpy() if hasattr(com, 'detach') else com
            gt3d.append(to_mm(labels.detach().cpu().numpy(), com_np, cube, nyu))
            pr3d.append(to_mm(pose.detach().cpu().numpy().reshape(labels.shape[0], -1), com_np, cube, nyu))
    hpe = HandposeEvaluation(np.concatenate(gt3d), np.concatenate(pr3d))
    n = hpe.err.shape[0]
    return hpe.getMeanError(), 100.0 * hpe.getNumFramesWithinMaxDist(40) / n
