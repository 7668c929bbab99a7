#!/bin/bash
# two copies of the probe at once (two processes on one GPU): producer kernel x consumer placement
for cfg in ${CFGS:-"conv:event" "conv_x3:event" "conv64:event" "conv:same" "fill:event"}; do
  python tools/r5_event_visibility.py ${N:-300} ${cfg%%:*} ${cfg#*:} 2>/dev/null | tail -${TAILN:-1} & P1=$!
  python tools/r5_event_visibility.py ${N:-300} ${cfg%%:*} ${cfg#*:} 2>/dev/null | tail -${TAILN:-1} & P2=$!
  wait $P1 $P2
done
echo "-- alone:"; python tools/r5_event_visibility.py ${N:-300} conv event 2>/dev/null | grep "stale$"
