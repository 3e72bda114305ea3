"""Comparison of two free-running detection lists (HIP pipeline vs CPU oracle).

The predict path ends in a score sort and a `max_per_img` cut: two candidates whose oracle scores agree to within fp32
noise may legitimately come out in the other order, or straddle the cut, when any upstream sum is associated
differently.  The stage-wise tests (which feed both sides the same tensors) are the bit-exactness gates; the end-to-end
tests use this matcher: every oracle detection must have a HIP detection with the same label and box, except
entries that tie with a neighbour / the cut-off score to within `tie`."""
import torch


def match_detections(p_boxes, p_scores, p_labels, r_boxes, r_scores, r_labels, box_tol=1e-2, tie=5e-5, max_odd=4,
                     score_tol=1e-4):
    p_boxes, p_scores, p_labels = p_boxes.detach().float().cpu(), p_scores.detach().float().cpu(), p_labels.detach().cpu()
    r_boxes, r_scores, r_labels = r_boxes.detach().float().cpu(), r_scores.detach().float().cpu(), r_labels.detach().cpu()
    k = r_scores.shape[0]
    assert p_scores.shape[0] == k, (p_scores.shape, k)
    if k == 0:
        return []
    cut = float(r_scores.min())
    gap = (r_scores[:, None] - r_scores[None, :]).abs() + torch.eye(k) * 1e9
    tied = (gap.min(1).values < tie) | ((r_scores - cut).abs() < tie)          # oracle entries that may move
    used = torch.zeros(p_scores.shape[0], dtype=torch.bool)
    pairs, odd = [], 0
    for j in range(k):
        d = (p_boxes - r_boxes[j]).abs().amax(1)
        d[(p_labels != r_labels[j]) | used] = float('inf')
        i = int(d.argmin())
        if float(d[i]) < box_tol and abs(float(p_scores[i]) - float(r_scores[j])) < score_tol:
            used[i] = True
            pairs.append((i, j))
            if i != j:
                assert bool(tied[j]), f'detection {j} moved to rank {i} without a score tie'
                odd += 1
        else:
            assert bool(tied[j]), f'oracle detection {j} (score {float(r_scores[j]):.6f}) has no counterpart'
            odd += 1
    assert odd <= max_odd, f'{odd} detections differ from the oracle'
    return pairs
