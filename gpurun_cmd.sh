bash tools/profile.sh r01
