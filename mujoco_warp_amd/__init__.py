"""mujoco_warp_amd: MI355X-native batched MuJoCo stepping engine behind the mujoco_warp Python API.

Public surface mirrors /root/reference/mujoco_warp/__init__.py:26-123 for the mj_step hot path
(put_model / put_data / make_data / get_data_into / reset_data / step / forward + stage functions + enums).
Physics runs in hand-written HIP kernels (csrc/) reached through the C ABI in include/mjhip.h.
"""

from . import mjcf
from .device import DeviceArray
from .device import copy
from .forward import KERNEL_NAMES
from .forward import StepGraph
from .forward import com_pos
from .forward import contact_force
from .forward import energy_pos
from .forward import energy_vel
from .forward import jac
from .forward import sensor_acc
from .forward import sensor_pos
from .forward import sensor_vel
from .forward import island
from .forward import sensor
from .forward import sleep
from .forward import update_sleep
from .forward import wake
from .forward import wake_collision
from .forward import wake_equality
from .forward import com_vel
from .forward import collision
from .forward import crb
from .forward import ctrl_noise
from .forward import efc_J_sparse
from .forward import qLD_dense
from .forward import euler
from .forward import factor_m
from .forward import forward
from .forward import fwd_acceleration
from .forward import fwd_actuation
from .forward import fwd_position
from .forward import fwd_velocity
from .forward import implicit
from .forward import kinematics
from .forward import make_constraint
from .forward import mul_m
from .forward import passive
from .forward import fwd_kinematics
from .forward import rne
from .forward import rungekutta4
from .forward import step1
from .forward import step2
from .forward import solve
from .forward import solve_m
from .forward import step
from .forward import timed_steps
from .forward import transmission
from .io import get_data_into
from .io import get_state
from .io import set_state
from .io import state_size
from .io import load_trajectory
from .io import make_data
from .io import override_model
from .io import put_data
from .io import put_model
from .io import reset_data
from .io import reset_data_keyframe
from .mjcf import MjData
from .mjcf import MjModel
from .mjcf import mj_resetDataKeyframe
from .types import BiasType
from .types import BroadphaseFilter
from .types import BroadphaseType
from .types import ConeType
from .types import Constraint
from .types import ConstraintState
from .types import ConstraintType
from .types import Contact
from .types import ContactType
from .types import Data
from .types import DisableBit
from .types import DynType
from .types import EnableBit
from .types import GainType
from .types import GeomType
from .types import IntegratorType
from .types import JointType
from .types import Model
from .types import ObjType
from .types import Option
from .types import OverflowType
from .types import SleepPolicy
from .types import SleepState
from .types import SolverType
from .types import State
from .types import Statistic
from .types import TrnType

__version__ = "0.1.0"
