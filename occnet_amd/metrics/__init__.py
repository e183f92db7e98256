"""Evaluation metric of the occupancy-and-flow challenge on the MI355X ray caster."""
from .ray_metrics import (calc_metrics, generate_lidar_rays, main, occ_class_names,  # noqa: F401
                          flow_class_names, process_one_sample)
