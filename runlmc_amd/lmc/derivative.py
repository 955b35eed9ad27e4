"""dL/dtheta = (alpha^T dK alpha - tr(K^-1 dK)) / 2 (mirror of reference
runlmc/lmc/derivative.py:4-12)."""


class Derivative:
    def derivative(self, dKdt):
        return 0.5 * (self.d_normal_quadratic(dKdt) - self.d_logdet_K(dKdt))

    def d_normal_quadratic(self, dKdt):
        raise NotImplementedError

    def d_logdet_K(self, dKdt):
        raise NotImplementedError
