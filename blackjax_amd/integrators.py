"""Integrator selectors.  Only velocity Verlet (coefficients [0.5, 1.0, 0.5],
blackjax/mcmc/integrators.py:321-322) has a HIP implementation; the other
palindromic integrators of the reference are out of scope."""


class _VelocityVerlet:
    coefficients = (0.5, 1.0, 0.5)

    def __repr__(self):
        return "velocity_verlet"


velocity_verlet = _VelocityVerlet()


def check_supported(integrator):
    if integrator is not velocity_verlet:
        raise NotImplementedError(
            "blackjax_amd implements the velocity_verlet integrator only; got %r" % (integrator,)
        )
