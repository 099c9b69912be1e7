#!/bin/bash
# What the GPU box gives one lease on the host side: cores by affinity, cgroup CPU quota, memory, masks.
echo "nproc: $(nproc)  nproc --all: $(nproc --all)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
echo "HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES CUDA_VISIBLE_DEVICES=$CUDA_VISIBLE_DEVICES"
grep -i "cpus_allowed_list" /proc/self/status
free -g | head -2
