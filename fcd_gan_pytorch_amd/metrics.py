"""On-device accuracy assessment.

The reference thresholds the density map on the GPU, then copies every sample to the
host and builds the 2x2 confusion matrix with NumPy (Demo_RSSS.py:345-354,
metrics.py:60-82) -- N device-to-host copies per batch.  Here the thresholded map, the
TP/FP/FN colour coding (CommonFunc.py:59-75) and the confusion counts stay on the
device; one 4-element tensor per batch is accumulated (and all-reduced across ranks
under data parallelism) and the derived scores use the reference's formulas
(metrics.py:11-58) on the host at epoch end.
"""
import numpy as np
import torch
import torch.distributed as dist


def threshold_map(cmap, prob_thresh=0.5):
    """cmask[cmap > prob_thresh] = 1 (Demo_RSSS.py:345-346), as a float {0,1} map."""
    return (cmap > prob_thresh).to(cmap.dtype)


def changemap_codes(change_mask, ref_mask, write_color=True, ref_map=(0, 1), dt_map=(0, 1)):
    """write_changemap_gdal (CommonFunc.py:59-75) on device: 0 background, 1 miss, 2 false
    detection, 3 true detection (write_color) or 0/1 detection otherwise.  Inputs (..,H,W)."""
    out = torch.zeros_like(change_mask, dtype=torch.float32)
    if write_color:
        out[(change_mask == dt_map[0]) & (ref_mask == ref_map[1])] = 1
        out[(change_mask == dt_map[1]) & (ref_mask == ref_map[0])] = 2
        out[(change_mask == dt_map[1]) & (ref_mask == ref_map[1])] = 3
    else:
        out[change_mask == dt_map[1]] = 1
    return out


class Evaluator:
    """Drop-in for metrics.Evaluator (num_class = 2) whose accumulation runs on the device.

    ``add_batch_map(gt, pre, gt_map, pre_map)`` accepts device tensors of any matching shape
    (a whole batch at once) and an optional ``valid`` mask (the reference scores only the
    owned centre of each patch, Demo_RSSS.py:351-354); NumPy inputs are accepted too."""

    def __init__(self, num_class=2, group=None):
        if num_class != 2:
            raise ValueError('the FCD-GAN demos use a 2-class evaluator')
        self.num_class = num_class
        self.group = group
        self._dev_counts = None
        self.confusion_matrix = np.zeros((2, 2))

    def reset(self):
        self._dev_counts = None
        self.confusion_matrix = np.zeros((2, 2))

    def add_batch_map(self, gt_image, pre_image, gt_map=(0, 1), pre_map=(0, 1), valid=None):
        gt, pre = torch.as_tensor(gt_image), torch.as_tensor(pre_image)
        assert gt.shape == pre.shape
        assert len(gt_map) == len(pre_map) == self.num_class
        if pre.device != gt.device:
            gt = gt.to(pre.device)
        cells = []
        for i in range(2):
            for j in range(2):
                m = (gt == gt_map[i]) & (pre == pre_map[j])
                if valid is not None:
                    m = m & valid
                cells.append(m.sum())
        c = torch.stack(cells).to(torch.int64)
        self._dev_counts = c if self._dev_counts is None else self._dev_counts + c

    def _sync(self):
        if self._dev_counts is not None:
            c = self._dev_counts
            from . import dp
            if dp.exchanging(self.group):
                c = c.clone()
                dist.all_reduce(c, group=self.group)
            self.confusion_matrix = self.confusion_matrix + c.cpu().numpy().reshape(2, 2).astype(np.float64)
            self._dev_counts = None
        return self.confusion_matrix

    # ---- derived scores: formulas of metrics.py:11-58 --------------------------------
    def Pixel_Accuracy(self):
        cm = self._sync()
        return np.diag(cm).sum() / cm.sum()

    def Pixel_Kappa(self):
        cm = self._sync()
        po = self.Pixel_Accuracy()
        pe = np.dot(cm.sum(axis=0), cm.sum(axis=1)) / np.square(cm.sum())
        return (po - pe) / (1 - pe)

    def Pixel_Accuracy_Class(self):
        cm = self._sync()
        acc = np.diag(cm) / cm.sum(axis=1)
        return np.nanmean(acc), acc

    def Pixel_Precision_Rate(self):
        cm = self._sync()
        return cm[1, 1] / (cm[0, 1] + cm[1, 1])

    def Pixel_Recall_Rate(self):
        cm = self._sync()
        return cm[1, 1] / (cm[1, 0] + cm[1, 1])

    def Pixel_F1_score(self):
        rec, pre = self.Pixel_Recall_Rate(), self.Pixel_Precision_Rate()
        return 2 * rec * pre / (rec + pre)

    def Mean_Intersection_over_Union(self):
        cm = self._sync()
        iou = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        ciou = iou[1].copy()
        return np.nanmean(iou), ciou

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self._sync()
        freq = np.sum(cm, axis=1) / np.sum(cm)
        iu = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        return (freq[freq > 0] * iu[freq > 0]).sum()
