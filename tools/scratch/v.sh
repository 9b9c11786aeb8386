for v in "" _a; do
echo "variant=$v"; LANEFRONT_LIBRARY=$PWD/lane_slam_amd/liblanefront$v.so python -m pytest tests -m gpu -q 2>&1 | tail -2
done
