import sys, runpy
sys.path.insert(0, '.')
import dart_planner_amd.capi as c
c._TYPED_API.pop('solve', None)
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
