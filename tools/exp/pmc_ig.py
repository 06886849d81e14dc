import sqlite3, sys
for f in sys.argv[1:]:
    db = sqlite3.connect(f)
    rows = list(db.execute("select kernel_name, grid_size, counter_name, avg(value) from counters_collection where kernel_name like '%ig_conv%' group by kernel_name, grid_size, counter_name"))
    cur = None
    for k, g, c, v in rows:
        key = (k.split('ig_conv_kernel')[1][:18], g)
        if key != cur: print(key); cur = key
        print("    %-32s %14.0f" % (c, v))
