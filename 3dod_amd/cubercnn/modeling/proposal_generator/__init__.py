from .rpn import RPNWithIgnore, RPN, StandardRPNHead, build_proposal_generator
