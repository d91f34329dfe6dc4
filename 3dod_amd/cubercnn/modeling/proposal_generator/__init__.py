from .rpn import RPNWithIgnore, RPN, StandardRPNHead, subsample_labels, matched_pairwise_iou, build_proposal_generator
