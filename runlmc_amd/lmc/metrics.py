"""Per-step diagnostics container (mirror of reference
runlmc/lmc/metrics.py:4-10)."""


class Metrics:
    def __init__(self):
        self.iterations = []
        self.grad_norms = []
        self.grad_error = []
        self.solv_error = []
        self.log_likely = []
