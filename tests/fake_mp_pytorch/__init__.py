"""
TEST INFRASTRUCTURE: an mp_pytorch-SHAPED facade over oracle/mp_oracle.py (``phase_gn`` / ``basis_gn`` / ``mp`` with the
class names and the call surface the reference uses: fancy_gym/black_box/factory/*.py, black_box_wrapper.py:57-125).

It exists for ONE purpose: to run tools/pin_against_mp_pytorch.py end to end where the real package cannot be installed
(``--package tests.fake_mp_pytorch``), so that the script's plumbing -- constructor kwargs, the call sequence, the switch
detection, the fixture format -- is tested.  It is the oracle wearing mp_pytorch's interface: outputs produced through it
pin NOTHING and are labelled with this package name.  ``BEHAVIOUR`` selects which setting of each "(?)" switch the facade
exhibits, so a test can check that the script reports exactly that setting.
"""
__version__ = "0.0-facade-over-the-oracle"

# the behaviour the facade exhibits for the four "(?)" switches (same names as oracle.TrajCfg / BasisCfg fields)
BEHAVIOUR = dict(relative_goal_mode="before_scale", goal_offset_mode="ignore", single_rbf_mode="unit_gap",
                 dmp_first_sample="init")
