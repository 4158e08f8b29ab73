"""
Environment ids that live on the hot path: the reference's NumPy reacher (the only inner env that needs no physics
engine), registered under the reference's ids (fancy_gym/envs/__init__.py:37-56) together
with its fancy_ProMP / fancy_DMP / fancy_ProDMP versions.
"""
from .registry import register

for _id, _links in (("fancy/SimpleReacher-v0", 2), ("fancy/LongSimpleReacher-v0", 5)):
    register(id=_id, entry_point="fancy_gym_amd.envs.classic_control.simple_reacher:SimpleReacherEnv",
             mp_wrapper="fancy_gym_amd.envs.classic_control.simple_reacher:SimpleReacherMPWrapper",
             max_episode_steps=200, kwargs={"n_links": _links})
