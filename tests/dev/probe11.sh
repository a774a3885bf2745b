cd /root/repo
make -C femus_amd/csrc/adapters > /dev/null 2>&1
g++ -O1 -std=c++17 -Ifemus_amd/csrc/adapters/mirror -Ifemus_amd/csrc/adapters -Iinclude tests/cpp/navier_stokes_adapters.cpp -o /tmp/nsa -Lfemus_amd/lib -lfemus_hip_adapters -lfemus_hip -Wl,-rpath,$PWD/femus_amd/lib
echo "--- 4 levels, application smoother settings (blocks 0,4; GMRES level solver; ILU; PREONLY x 2), stabilised Q1/Q1"
timeout 600 /tmp/nsa 10 4 0.0 /tmp/o.bin 0 4 0 1 0 1 2>&1 | grep -E "Nonlinear iteration|newton steps" | tail -12
echo "--- same, flexible GMRES outer"
timeout 600 /tmp/nsa 10 4 0.0 /tmp/o2.bin 0 4 0 2 0 1 2>&1 | grep -E "Nonlinear iteration|newton steps" | tail -6
